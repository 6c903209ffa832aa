#!/usr/bin/env python3
"""Static instruction / register statistics per kernel from the gfx950 assembly of trp_kernels.hip."""
import re
import subprocess
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/isa"
os.makedirs(out, exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=off",
                "-I" + ROOT + "/include", "--save-temps=obj", "-c", ROOT + "/rust-triplets_amd/csrc/trp_kernels.hip",
                "-o", out + "/k.o"], check=True, stderr=subprocess.DEVNULL)
s = open(out + "/trp_kernels-hip-amdgcn-amd-amdhsa-gfx950.s").read()
for f in re.split(r'\n\s*\.globl\s+', s)[1:]:
    name = f.split('\n', 1)[0].split(';')[0].strip()
    lines = [l.strip() for l in f.split('\n')]
    ins = [l for l in lines if l and not l.startswith(('.', ';', '_')) and not l.endswith(':')]
    cnt = lambda pre: sum(1 for l in ins if l.startswith(pre))
    g = lambda pat: (re.search(pat, f) or [None, None])[1]
    short = re.sub(r'^_ZN\d+_GLOBAL__N_1\d*', '', name)[:28]
    if len(sys.argv) > 1 and not any(a in name for a in sys.argv[1:]):
        continue
    print("%-28s valu %5d salu %5d ds %4d mem %4d | vgpr %s scratch %s occ %s lds %s" % (
        short, cnt('v_'), cnt('s_'), cnt('ds_'), cnt(('global_', 'buffer_', 'flat_', 'scratch_')),
        g(r'; NumVgprs: (\d+)'), g(r'; ScratchSize: (\d+)'), g(r'; Occupancy: (\d+)'), g(r'; LDSByteSize: (\d+)')))
