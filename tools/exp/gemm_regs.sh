#!/bin/bash
# registers / scratch / in-loop full drains of every gemm256p instantiation: tools/exp/gemm_regs.sh [extra -D flags]
cd "$(dirname "$0")/../.."
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=fast -Wno-unused-result -fno-gpu-rdc -mllvm -amdgpu-early-inline-all=true -mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc $FLAGS "$@" --cuda-device-only -S devias_amd/csrc/gemm.hip -o /tmp/gemm_regs.s || exit 1
python3 - <<'PY'
import re
L=open('/tmp/gemm_regs.s').read().split('\n')
for i,l in enumerate(L):
    m=re.match(r'^(_ZN\S*gemm256p_kernel\S*):',l)
    if not m: continue
    n=m.group(1); j=i
    while 's_endpgm' not in L[j]: j+=1
    body=L[i:j+1]
    meta={k:v for x in L if n in x for k,v in re.findall(r'\.(num_vgpr|private_seg_size|num_agpr|sgpr_count), (\d+)',x)} if False else {}
    vg=[x for x in L if x.strip().startswith('.set '+n+'.num_vgpr')]
    sc=[x for x in L if x.strip().startswith('.set '+n+'.private_seg_size')]
    print(n[-40:], 'lines',len(body),'vmcnt(0):',sum('vmcnt(0)' in b for b in body),'scratch:',sum('scratch_' in b for b in body), vg[0].split(',')[-1] if vg else '', sc[0].split(',')[-1] if sc else '')
PY
