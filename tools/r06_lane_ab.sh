#!/bin/bash
# A/B of lane-kernel variant libraries (tools/lane_variants.py -> tools/.prof/var_*.so) on the default path (lane kernel to the
# cap + hand-off) and on the pure lane kernel.  usage (through gpurun): bash tools/r06_lane_ab.sh tag "v1 v2 ..." [cases]
set -u
tag=${1:-r06_lane_ab}
vars=${2:-"base new"}
cases=${3:-10:32768,20:65536,10:65536}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
f=$out/lane_ab.txt
: > $f
for v in $vars; do
  echo "== $v (default path)" >> $f
  QMPC_LIB=$root/tools/.prof/var_$v.so timeout 600 python tools/handoff_bench.py --cases $cases --caps default --reps 8 2>&1 | grep -v amdgpu.ids >> $f
done
for v in $vars; do
  echo "== $v (pure lane kernel, QMPC_VARIANT=4)" >> $f
  QMPC_LIB=$root/tools/.prof/var_$v.so timeout 600 python tools/lane_bench.py --skip-wave --reps 5 --sample 64 --cases $cases 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('N=%d B=%d %.3f ms  iters mean/max %s  err_vs_oracle %.1e  oracle_iters_equal %.4f'%(r['N'],r['B'],r['lane_ms'],r['lane_iters_mean_max'],r['lane_vs_oracle_linf'],r['oracle_iters_equal_frac']))" >> $f
done
cat $f
