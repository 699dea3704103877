#!/bin/bash
# A/B of two lane-kernel variant libraries (tools/lane_variants.py) with the bit comparisons that every change of the passes has
# to survive: tools/lane_bits.py (converged mode: pairs, full wavefronts, warm start, the other models) and
# tools/refmode_pair_bits.py (reference mode; shards against their blocks of a full launch), then tools/r06_lane_ab.sh.
# usage (through gpurun): bash tools/r06_variant_ab.sh <old> <new> [cases]
set -u
old=${1:?old variant}; new=${2:?new variant}; cases=${3:-10:32768,10:16384,20:32768}
root=$(pwd); o=/tmp/r06_variant_ab; mkdir -p $o gpurun_out/$new
QMPC_LIB=$root/tools/.prof/var_$new.so timeout 600 python tools/lane_bits.py $o/bits.npz > $o/bits.log 2>&1
QMPC_LIB=$root/tools/.prof/var_$new.so timeout 600 python tools/refmode_pair_bits.py $o/rbits.npz > gpurun_out/$new/rbits_self.txt 2>&1
QMPC_LIB=$root/tools/.prof/var_$old.so timeout 600 python tools/lane_bits.py --compare $o/bits.npz > gpurun_out/$new/bits_cmp.txt 2>&1
QMPC_LIB=$root/tools/.prof/var_$old.so timeout 600 python tools/refmode_pair_bits.py --compare $o/rbits.npz > gpurun_out/$new/rbits_cmp.txt 2>&1
echo "bit-identical arrays: $(grep -c identical gpurun_out/$new/bits_cmp.txt) converged, $(grep -c identical gpurun_out/$new/rbits_cmp.txt) reference mode"
grep DIFF gpurun_out/$new/bits_cmp.txt gpurun_out/$new/rbits_cmp.txt
grep shard gpurun_out/$new/rbits_self.txt
bash tools/r06_lane_ab.sh $new "$old $new" $cases | tail -40
