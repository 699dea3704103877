#!/bin/bash
# Round-6 counter passes over ONE bench.py leg (rocprofv3 --pmc only, one counter set per run, as the microarchitecture
# guide prescribes): SQ issue / wait / MFMA counters, FP64 instruction mix, FETCH_SIZE, WRITE_SIZE + L2 hits.
# usage through gpurun:  bash tools/pmc_r06.sh <tag> <batch> <horizon>     -> gpurun_out/<tag>/ + profiles-ready JSONs
set -u
tag=$1; B=$2; N=$3
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
declare -A SETS
SETS[a]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
SETS[b]="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
SETS[c]="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC"
SETS[d]="FETCH_SIZE"
SETS[e]="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
steps=6; [ "$B" -ge 16384 ] && steps=3
for p in a b c d e; do
  timeout 600 rocprofv3 --pmc ${SETS[$p]} -f csv -d "$out/pmc_$p" -- python "$root/bench.py" --no-cpu-baseline --no-in-flight --no-large-batch \
      --no-closed-loop --no-reference-mode --batch $B --horizon $N --steps $steps --warmup 2 > "$out/pmc_$p.json" 2> "$out/pmc_$p.err"
done
cd "$root"
python tools/pmc_r06_summary.py "$out" $B $N | tee "$out/summary.txt"
find "$out" -name "*.csv" -size +4M -delete
