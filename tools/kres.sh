#!/bin/bash
# usage: tools/kres.sh file.hip [extra flags] -- compile for gfx950 and print per-kernel resource usage
f=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$f" -o /tmp/$(basename "$f").o "$@" -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
name=None; d={}
for l in sys.stdin:
    if 'error' in l or 'warning' in l: print(l.rstrip())
    m=re.search(r'Function Name: (\S+)',l)
    if m: name=m.group(1); d={}; continue
    for k in ('TotalSGPRs','VGPRs','AGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','LDS Size \[bytes/block\]'):
        m=re.search(k+r': (\d+)',l)
        if m: d[k.split(' ')[0]]=m.group(1)
    if 'LDS Size' in l:
        import subprocess
        print(subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()[:70], d)
"
