#!/bin/bash
# SQ counter passes (8 counters per pass, rocprofv3 --pmc only) over bench.py; usage through gpurun:
#   bash tools/pmc_sq.sh <tag> [bench args...]   -> gpurun_out/<tag>/sq_{a,b,c}/ + sq_summary.json
set -u
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
B="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
C="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC"
for p in a b c; do
  case $p in a) set_="$A";; b) set_="$B";; c) set_="$C";; esac
  timeout 600 rocprofv3 --pmc $set_ -f csv -d "$out/sq_$p" -- python "$root/bench.py" --no-cpu-baseline --no-in-flight --no-large-batch --no-closed-loop --no-reference-mode --steps 6 --warmup 2 "$@" > "$out/sq_$p.json" 2> "$out/sq_$p.err"
done
cd "$root"
python tools/pmc_sq_summary.py "$out" > "$out/sq_summary.json"
cat "$out/sq_summary.json"
find "$out" -name "*.csv" -size +4M -delete
