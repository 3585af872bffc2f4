#!/bin/bash
# Per-kernel register / LDS / scratch table of the gfx950 code object (hipcc --save-temps of limo_hip.hip).
# usage: [ISA_SRC=depth.hip] scripts/isa_resources.sh [extra hipcc flags]   -> prints name vgpr agpr sgpr lds scratch occupancy
set -e
D=${ISA_DIR:-/tmp/isa}
mkdir -p $D && cd $D
SRC=${ISA_SRC:-limo_hip.hip}
export ISA_BASE=${SRC%.*}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c --save-temps -Wno-unused-function "$@" /root/repo/limo_amd/csrc/$SRC -o $ISA_BASE.o
python3 - <<'PY'
import os,re,subprocess
s=open('%s-hip-amdgcn-amd-amdhsa-gfx950.s' % os.environ['ISA_BASE']).read()
# metadata blocks
for m in re.finditer(r'\.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)', s, re.S):
    ag,lds,name,scr,sg,vg=m.groups()
    name=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip().split('(')[0]
    tot=int(vg)
    alloc=(tot+7)//8*8
    occ=min(8,512//alloc) if alloc else 8
    print("%-34s vgpr+agpr %3s (agpr %3s) sgpr %3s lds %6s scratch %3s -> %d waves/SIMD"%(name[-34:],vg,ag,sg,lds,scr,occ))
PY
