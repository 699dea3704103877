#!/bin/bash
# Round 6, review item 1(a): what can ANY byte saving buy the lane kernel?  Diagnostic builds of the lane unit
# (tools/lane_variants.py: dbase = 16 fixed rounds, dalias = + every wavefront of an XCD in ONE workspace block (all L2 hits),
# dnost = + pass stores elided, dboth = both) timed on the two large-batch legs, pure lane kernel (QMPC_VARIANT=4, no hand-off).
# Also: the footprint sweep with the phase-counter build (cycles per knot-iteration at 32 ... 1024 resident wavefronts).
# usage (through gpurun): bash tools/r06_lane_traffic.sh [tag] -> gpurun_out/<tag>/
set -u
tag=${1:-r06_lane_traffic}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
f=$out/lane_traffic_bound.txt
echo "# lane kernel, fixed 16 rounds, pure lane kernel (QMPC_VARIANT=4): kernel ms (median of 5)" > $f
for v in base dbase dalias dnost dboth stlow; do
  so=$root/tools/.prof/var_$v.so
  [ -f $so ] || continue
  for c in 10:32768 20:65536 10:65536; do
    r=$(QMPC_LIB=$so QMPC_LANE_CAP=0 timeout 300 python tools/lane_bench.py --skip-wave --reps 5 --sample 4 --cases $c 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('N=%d B=%d %.3f ms  iters mean/max %s  err_vs_oracle %.1e'%(r['N'],r['B'],r['lane_ms'],r['lane_iters_mean_max'],r['lane_vs_oracle_linf']))")
    echo "$v: $r" >> $f
  done
done
echo "# default path (lane to cap + hand-off), production build vs stores of the partner lanes predicated off (stlow)" >> $f
for v in base stlow; do
  echo "== $v" >> $f
  QMPC_LIB=$root/tools/.prof/var_$v.so timeout 600 python tools/handoff_bench.py --cases 10:32768,20:65536,10:65536 --caps default --reps 8 >> $f 2>&1
done
g=$out/lane_footprint_sweep.txt
echo "# phase-counter build (QL_PROFILE), pure lane kernel: cycles per knot-iteration vs resident wavefronts" > $g
for c in 10:1024 10:2048 10:4096 10:8192 10:16384 10:32768 10:65536 20:2048 20:65536; do
  QMPC_LIB=$root/tools/.prof/libqmpc_hip.so QMPC_LANE_CAP=0 timeout 300 python tools/lane_bench.py --skip-wave --reps 1 --sample 4 --cases $c 2>&1 | awk '/^lane profile/{n++} n==1 && !/^{/' >> $g
done
cat $f
cat $g
