#!/bin/bash
# VGPRs / spills / occupancy of every k_conv_sk instance:  bash tools/kernel_regs.sh [file.hip]
F=${1:-agile3d_amd/csrc/spconv.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I agile3d_amd/csrc -I include -c $F -o /tmp/kr.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re,sys
cur=None
for line in sys.stdin:
    m=re.search(r"Function Name: (\S+)",line)
    if m: cur={"n":m.group(1)}; continue
    if cur is None: continue
    for key,pat in (("v",r" VGPRs: (\d+)"),("occ",r"Occupancy \[waves/SIMD\]: (\d+)"),("ss",r"SGPRs Spill: (\d+)"),("vs",r"VGPRs Spill: (\d+)")):
        m=re.search(pat,line)
        if m: cur[key]=int(m.group(1))
    if "vs" in cur:
        m=re.search(r"k_conv_skILi(\d+)ELi(\d+)ELi(\d+)ELb(\d)",cur["n"])
        if m: print("k_conv_sk<%s,%s,%s,%s> vgpr=%d occ=%d sgpr_spill=%d vgpr_spill=%d"%(m.group(1),m.group(2),m.group(3),m.group(4),cur["v"],cur["occ"],cur["ss"],cur["vs"]))
        cur=None
'
