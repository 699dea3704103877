#!/usr/bin/env python3
"""Instruction mix per basic block of a kernel in an AMDGPU assembly listing (hipcc -S --cuda-device-only).
  python tools/isa_blocks.py file.s kernel-name-substring [min-block-size]"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 40
start = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and key in l][0]
ends = [i for i, l in enumerate(lines) if i > start and ".end_amdhsa_kernel" in l]; end = ends[0] if ends else len(lines)
body = lines[start:end]
blocks, cur, name = [], [], 'entry'
for l in body:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blocks.append((name, cur)); cur = []; name = m.group(1)
    else:
        t = l.strip()
        if t and not t.startswith(';') and not t.startswith('.'):
            cur.append(t)
blocks.append((name, cur))
print("instructions:", sum(len(b) for _, b in blocks))
for n, b in blocks:
    if len(b) >= minsz:
        ops = {}
        for t in b:
            o = t.split()[0]
            k = ('mfma' if 'mfma' in o else 'ds_read' if o.startswith('ds_read') else 'ds_write' if o.startswith('ds_write')
                 else 'ds_bperm' if 'permute' in o else 'readlane' if 'readlane' in o else 'permlane' if 'permlane' in o
                 else 'dpp' if 'dpp' in t else 'f64' if 'f64' in o else 'waitcnt' if 'waitcnt' in o
                 else 'nop' if 's_nop' in o else 's_' if o.startswith('s_') else 'v_other')
            ops[k] = ops.get(k, 0) + 1
        print(n, len(b), dict(sorted(ops.items())))
