#!/bin/bash
# usage: tools/kernel_resources.sh jrender_amd/csrc/file.hip  -> per-kernel VGPR / SGPR / scratch / occupancy / LDS
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -x hip $KR_FLAGS -c "$1" -o /dev/null --cuda-device-only -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur={}
for l in sys.stdin:
    m=re.search(r'remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (.*?) \[-R',l)
    if not m: continue
    k,v=m.group(1),m.group(2)
    if k=='Function Name':
        cur={'n':v}
    cur[k]=v
    if k.startswith('LDS'):
        print(cur['n'][:60], 'V',cur.get('VGPRs'),'S',cur.get('TotalSGPRs'),'scr',cur.get('ScratchSize [bytes/lane]'),'occ',cur.get('Occupancy [waves/SIMD]'),'lds',v)
"
