#!/bin/bash
# Registers, scratch and LDS of every kernel in a .hip source (compiler remarks; no GPU needed).
# usage: tools/kernel_resources.sh query-engine_amd/csrc/qe_static.hip
cd "$(dirname "$0")/../query-engine_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -x hip -c "$(basename "$1")" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|Occupancy|ScratchSize|LDS Size" | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | paste - - - - - - | sed -E 's/Function Name: _ZN2qe[0-9]*//' | cut -c1-200
