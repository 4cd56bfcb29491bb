#!/bin/bash
# VGPRs / spills / LDS of the gfx950 kernels of one translation unit's object: tools/kernel_regs.sh pcr_gicp [name-filter]
U=$1; F=${2:-.}
D=$(dirname "$0")/../point-cloud-registration-with-global-refinement_amd/csrc
objcopy -O binary --only-section=.hip_fatbin $D/$U.o /tmp/$U.fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=/tmp/$U.fat.bin --output=/tmp/$U.gfx950.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/$U.gfx950.co | python3 -c "
import sys,re
name=None; rec={}
for line in sys.stdin:
    m=re.match(r'\s+\.(name|vgpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):\s+(\S+)', line)
    if m:
        rec[m.group(1)]=m.group(2)
    if line.strip().startswith('- .agpr_count') or line.strip().startswith('- .args'):
        if rec.get('name'): print(rec)
        rec={}
if rec.get('name'): print(rec)
" | c++filt | grep -E "$F"
