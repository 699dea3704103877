#!/bin/bash
# run tools/lane_bench.py (lane kernel only) once per variant library in tools/.prof/var_*.so; print one line per case
cases=${1:-10:1024,10:32768,10:65536}
for v in tools/.prof/var_*.so; do
  n=$(basename $v .so)
  QMPC_LIB=$PWD/$v timeout 200 python tools/lane_bench.py --skip-wave --reps 5 --sample 16 --cases $cases 2>/dev/null | python3 -c "
import sys,json
out=[]
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); out.append('N%d B%d %.2fms %.2fM/s err %.0e'%(r['N'],r['B'],r['lane_ms'],r['lane_solves_per_s']/1e6,r['lane_vs_oracle_linf']))
print('$n', ' | '.join(out))"
done
