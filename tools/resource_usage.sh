#!/bin/bash
# Register / scratch / occupancy figures of every kernel of one .hip file (gfx950), one line each.
#   tools/resource_usage.sh sage-icp_amd/csrc/kernels.hip [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c "$f" -o /dev/null \
    -Rpass-analysis=kernel-resource-usage --cuda-device-only "$@" 2>&1 | grep "remark:" | sed 's/ \[-Rpass.*//' |
  awk '/Function Name:/ {name=$NF} /TotalSGPRs:/ {s=$NF} / VGPRs:/ {v=$NF} /AGPRs:/ {a=$NF}
       /ScratchSize/ {sc=$NF} /Occupancy/ {o=$NF} /LDS Size/ {print name, "vgpr", v, "agpr", a, "sgpr", s, "scratch", sc, "occ", o, "lds", $NF}' |
  while read n rest; do echo "$(echo $n | c++filt | sed 's/(.*//') $rest"; done
