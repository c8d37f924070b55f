#!/bin/bash
# VGPRs / spills / LDS / occupancy of every kernel (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel
cd "$(dirname "$0")/.."
for f in ${@:-raster sort projection ingest}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
    -fhip-fp32-correctly-rounded-divide-sqrt -Rpass-analysis=kernel-resource-usage \
    -c godotgaussiansplatting_amd/csrc/$f.hip -o /tmp/_kr_$f.o 2>&1 | python3 -c "
import sys,re,subprocess
cur=None
for line in sys.stdin:
    m=re.search(r'remark: (.*?) \[-Rpass', line)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith('Function Name:'):
        name=t.split(':',1)[1].strip()
        try: name=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt',name],capture_output=True,text=True).stdout.strip()
        except Exception: pass
        name=re.sub(r'\(.*','',name).replace('gsplat::(anonymous namespace)::','')
        m2=re.match(r'_ZN6gsplat12_GLOBAL__N_1(\d+)(.*)',name)
        if m2: name=m2.group(2)[:int(m2.group(1))]+re.sub(r'^(IL[bi]\d+E(?:Li\d+E)*)?.*',r'\1',m2.group(2)[int(m2.group(1)):])
        cur={'name':name}
    elif cur is not None:
        k,v=t.split(':',1); cur[k.strip()]=v.strip()
        if k.strip()=='LDS Size [bytes/block]':
            print(f\"{cur['name']:<44} VGPR {cur.get('VGPRs','?'):>4} AGPR {cur.get('AGPRs','?'):>3} spill {cur.get('VGPR Spill','?'):>3} scratch {cur.get('ScratchSize [bytes/lane]','?'):>4} occ {cur.get('Occupancy [waves/SIMD]','?'):>2} LDS {v.strip():>6}\")
"
done
