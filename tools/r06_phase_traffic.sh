#!/bin/bash
# Round 6, HISTORY_r06.md section 10: the traffic-bound experiment per PHASE on the final lane passes, and the clock under the
# lane kernel.  Variants first (container):
#   python tools/lane_variants.py pbase="-DQL_PROFILE -DQL_DIAG_ROUNDS=16" palias="-DQL_PROFILE -DQL_DIAG_ROUNDS=16 -DQL_DIAG_ALIAS=8" \
#          pnost="-DQL_PROFILE -DQL_DIAG_ROUNDS=16 -DQL_DIAG_NOSTORE"
# usage (through gpurun): bash tools/r06_phase_traffic.sh  -> gpurun_out/lane_prof_traffic.txt, gpurun_out/clocks_during.txt
mkdir -p gpurun_out
for v in pbase palias pnost; do for c in 10:32768 20:65536; do
  echo "== $v $c"
  QMPC_LIB=$PWD/tools/.prof/var_$v.so QMPC_LANE_CAP=0 timeout 300 python tools/lane_bench.py --skip-wave --reps 3 --sample 4 --cases $c 2>&1 | grep -E "^lane profile|^  [A-C]|lane_ms" | cut -c1-300
done; done > gpurun_out/lane_prof_traffic.txt 2>&1
# clocks and power while the production lane kernel (config 3) runs back to back
(for i in 1 2 3 4 5 6 7 8; do sleep 1.5; rocm-smi --showclocks 2>/dev/null | grep -iE "sclk|mclk" | head -3; rocm-smi --showpower 2>/dev/null | grep -i "power" | head -2; done) > gpurun_out/clocks_during.txt 2>&1 &
QMPC_VARIANT=4 QMPC_LANE_CAP=0 timeout 120 python tools/lane_bench.py --skip-wave --reps 300 --sample 4 --cases 20:65536 2>&1 | grep lane_ms | cut -c1-200
wait
cat gpurun_out/lane_prof_traffic.txt gpurun_out/clocks_during.txt
