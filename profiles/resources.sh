#!/bin/bash
# VGPR / scratch / occupancy of every kernel of a source file:  bash profiles/resources.sh chain.hip
cd "$(dirname "$0")/../bsms-gnn_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC ${EXTRA} -c $1 -o /tmp/res_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
rows=[]; cur=None
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur={'name':m.group(1)}; rows.append(cur)
    for k,kk in [('VGPRs','V'),('AGPRs','A'),(r'ScratchSize \[bytes/lane\]','scr'),(r'Occupancy \[waves/SIMD\]','occ')]:
        m=re.search(' '+k+r': (\d+)',line)
        if m and cur is not None: cur[kk]=int(m.group(1))
for r in rows:
    n=r['name']; n=re.sub(r'_ZN12_GLOBAL__N_1\d+','',n); n=re.sub(r'EEvN4bsms.*','',n)
    print('%-44s V %3d A %3d scratch %4d occ %d'%(n[:44],r.get('V',-1),r.get('A',-1),r.get('scr',-1),r.get('occ',-1)))
"
rm -f /tmp/res_$$.o
