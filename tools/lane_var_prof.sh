#!/bin/bash
# phase profiles (QL_PROFILE builds made by tools/lane_variants.py) of every variant library, first profile block per case
cases=${1:-10:1024}
for v in tools/.prof/var_*.so; do
  echo "== $(basename $v .so)"
  QMPC_LIB=$PWD/$v timeout 200 python tools/lane_bench.py --skip-wave --reps 1 --sample 4 --cases $cases 2>&1 | awk '/^lane profile/{n++} n==1 && !/^{/' 
done
