#!/bin/bash
# What the GPU box offers for sampling clocks and power without going through rocm-smi.
O=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/sysfs
mkdir -p $O
{
  for c in /sys/class/drm/card*; do
    [ -e $c/device/vendor ] || continue
    echo "== $c vendor $(cat $c/device/vendor 2>/dev/null) -> $(readlink -f $c/device)"
    ls $c/device | tr '\n' ' '
    echo
    for f in pp_dpm_sclk pp_dpm_mclk gpu_busy_percent mem_busy_percent current_link_speed; do
      [ -r $c/device/$f ] && { echo "-- $f"; cat $c/device/$f; }
    done
    for h in $c/device/hwmon/hwmon*; do
      echo "-- $h"; ls $h | tr '\n' ' '; echo
      for f in freq1_input freq2_input power1_average power1_input temp1_input temp2_input; do
        [ -r $h/$f ] && echo "$f $(cat $h/$f)"
      done
    done
  done
  echo "== rocm-smi timing"
  time rocm-smi --showclocks --showpower --json > /dev/null
  time rocm-smi -d 0 --showclocks --json
  echo "== amd-smi"; which amd-smi && (time amd-smi metric -g 0 --clock --power --json | head -60)
  python3 -c "import torch;print(torch.cuda.get_device_properties(0))"
} > $O/sysfs.txt 2>&1
tail -5 $O/sysfs.txt
