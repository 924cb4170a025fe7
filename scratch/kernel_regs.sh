#!/bin/bash
# usage: scratch/kernel_regs.sh <file.hip> [name filter]  -- register / LDS / scratch use of every kernel in a translation unit
set -e
src=$1; filt=${2:-.}
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I/root/repo/include -I/root/repo/lvt_amd/csrc --cuda-device-only -c "$src" -o $tmp/dev.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
  python3 -c "
import sys,re
cur=None
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur={'name':m.group(1)}; continue
    for key in ('VGPRs','AGPRs','ScratchSize','Occupancy','LDS Size','SGPRs'):
        m=re.search(key+r'[^:]*: (\d+)',line)
        if m and cur is not None: cur[key]=m.group(1)
    if 'LDS Size' in line and cur:
        print(cur['name'][:110], {k:v for k,v in cur.items() if k!='name'}); cur=None
" | grep -E "$filt" | c++filt | sort -u
rm -rf $tmp
