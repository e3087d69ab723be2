#!/bin/bash
# One session, one box: the read-stream ceiling probe and the headline kernel (tools/probe.py c5), alternated, so that
# the kernel's rate can be put beside the ceiling of that session.  Build the probe first, here, where hipcc is:
#   hipcc -O3 --offload-arch=gfx950 -o ab/stream_probe tools/experiments/stream_probe.hip
# (ab/ is git-ignored and travels to the GPU box with the tree.)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ceiling
mkdir -p $O; : > $O/log.txt
for i in 1 2 3; do
  echo "== stream $i" >> $O/log.txt
  timeout -k 10 120 $R/ab/stream_probe 256000000 2>&1 | grep -E "cols=4 nt=1|triad nt=1" >> $O/log.txt || exit 1
  echo "== headline $i" >> $O/log.txt
  timeout -k 10 120 python3 -u $R/tools/probe.py c5 256000000 5 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/log.txt || exit 1
done
cat $O/log.txt | cut -c1-330
