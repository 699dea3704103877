#!/usr/bin/env python3
"""Dump the kernel statistics of a rocprofv3 rocpd (.db) result as text.
usage: tools/rocprof_summary.py <results.db> [out.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
print("# rocprofv3 --kernel-trace --stats : per-kernel summary (name, calls, total us, avg us, percent)", file=out)
for name, calls, total, avg, pct in db.execute("select * from top_kernels"):
    print(f"{calls:6d} calls  total {total:12.3f} us  avg {avg:10.3f} us  {pct:6.2f} %  {name}", file=out)
print("# dispatch geometry of the solve kernel (grid, workgroup, LDS bytes, VGPRs incl. AGPRs, SGPRs)", file=out)
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = list(db.execute("select * from kernels where name like '%qmpc_solve_kernel%' limit 1"))
if rows:
    d = dict(zip(cols, rows[0]))
    keep = {k: v for k, v in d.items() if any(t in k.lower() for t in ("grid", "workgroup", "lds", "vgpr", "sgpr", "scratch", "duration"))}
    print(keep, file=out)
