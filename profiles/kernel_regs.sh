#!/bin/bash
# registers and scratch of the loop kernels in a built library: bash profiles/kernel_regs.sh [lib.so] [name pattern]
LIB=${1:-sage-icp_amd/libsageicp_hip.so}; PAT=${2:-k_loop}
T=$(mktemp -d); cp $LIB $T/lib.so; (cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null 2>&1
for f in lib.so.*gfx950; do /opt/rocm/lib/llvm/bin/llvm-readelf --notes $f 2>/dev/null | grep -E "\.name:|\.vgpr_count|\.private_segment_fixed_size" | paste - - - | grep "$PAT" | sed 's/  */ /g' | cut -c1-170; done); rm -rf $T
