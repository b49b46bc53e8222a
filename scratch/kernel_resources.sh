#!/bin/bash
# register / spill / occupancy table of one translation unit's kernels (hipcc -Rpass-analysis=kernel-resource-usage), e.g.
#   scratch/kernel_resources.sh gemm_h2.hip
cd /root/repo/tf-faster-rcnn_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -I/root/repo/include -I. -c "$1" -o /tmp/kr_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m: cur={"name":m.group(1)}; continue
    for k in ("VGPRs","AGPRs","SGPRs Spill","VGPRs Spill","Occupancy \[waves/SIMD\]","ScratchSize \[bytes/lane\]"):
        m=re.search(r"\s"+k+r": (\d+)",l)
        if m and cur is not None: cur[k]=m.group(1)
    if "LDS Size" in l and cur:
        print("%-70s vgpr %s agpr %s sspill %s vspill %s occ %s scratch %s" % (cur["name"][:70],cur.get("VGPRs"),cur.get("AGPRs"),cur.get("SGPRs Spill"),cur.get("VGPRs Spill"),cur.get("Occupancy \[waves/SIMD\]"),cur.get("ScratchSize \[bytes/lane\]"))); cur=None
'
rm -f /tmp/kr_$$.o
