#!/bin/bash
# usage: tools_resource_usage.sh file.hip  -> one line per kernel: name vgpr agpr spill scratch lds occupancy
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -c "$1" -o /tmp/_ru.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur={}
for line in sys.stdin:
    m=re.search(r'remark: [^:]+:\d+:\d+:\s+(.*?)\s*\[-Rpass',line)
    if not m: continue
    t=m.group(1)
    if t.startswith('Function Name:'):
        if cur: print(cur)
        name=t.split(':',1)[1].strip()
        try: name=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt',name],capture_output=True,text=True).stdout.strip()
        except Exception: pass
        cur={'name':name[:110]}
    else:
        k,v=t.split(':',1); cur[k.strip()]=v.strip()
if cur: print(cur)
"
