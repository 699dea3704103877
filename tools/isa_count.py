"""Instruction mix of one kernel in a hipcc -S listing, whole body and per basic-block range.
python tools/isa_count.py file.s kernel_substring"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(txt) if re.match(r'^_Z\w+:', l) and key in l)
end = next(i for i in range(start, len(txt)) if txt[i].startswith('.Lfunc_end'))
ops = collections.Counter()
blocks = []   # (label, counter)
cur = ('entry', collections.Counter())
for line in txt[start + 1:end]:
    s = line.strip()
    if not s or s.startswith(';'):
        continue
    if s.endswith(':') or re.match(r'^\.LBB\w+:', s):
        blocks.append(cur)
        cur = (s.split(':')[0], collections.Counter())
        continue
    if s.startswith('.'):
        continue
    op = s.split()[0]
    ops[op] += 1
    cur[1][op] += 1
blocks.append(cur)


def grp(c):
    g = collections.Counter()
    for op, n in c.items():
        if op.startswith('scratch_'): g['scratch'] += n
        elif op.startswith('global_load'): g['gload'] += n
        elif op.startswith('global_store'): g['gstore'] += n
        elif op.startswith('ds_'): g['lds'] += n
        elif op.startswith('v_accvgpr'): g['accvgpr'] += n
        elif 'f64' in op: g['f64'] += n
        elif op.startswith('v_'): g['v_other'] += n
        elif op.startswith('s_waitcnt'): g['waitcnt'] += n
        elif op.startswith('s_'): g['salu'] += n
        else: g[op] += n
    return g


print('total', sum(ops.values()), dict(grp(ops)))
for lab, c in blocks:
    n = sum(c.values())
    if n >= 200:
        print(f'{lab:>12} {n:6d}', dict(grp(c)))
vo = collections.Counter({op: n for op, n in ops.items() if op.startswith('v_') and 'f64' not in op and 'accvgpr' not in op})
print(vo.most_common(12))
