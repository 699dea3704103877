#!/bin/bash
# rocprofv3 --pmc passes over the lane-per-instance kernel (tools/lane_bench.py, lane only); through gpurun:
#   bash tools/pmc_lane.sh <tag> <case N:B> [model]   -> gpurun_out/<tag>/pmc_summary.json
set -u
tag=$1; case_=$2; model=${3:-go1}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
declare -A SETS
SETS[a]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
SETS[b]="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
SETS[c]="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC"
SETS[d]="FETCH_SIZE"
SETS[e]="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
SETS[f]="TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum GRBM_GUI_ACTIVE"
SETS[g]="SQ_IFETCH SQ_WAIT_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"
for p in a b c d e f g; do
  timeout 300 rocprofv3 --pmc ${SETS[$p]} -f csv -d "$out/pmc_$p" -- python "$root/tools/lane_bench.py" --skip-wave --reps 2 --sample 4 --cases $case_ --model $model > "$out/pmc_$p.json" 2> "$out/pmc_$p.err"
done
cd "$root"
python - "$out" <<'PY' | tee "$out/pmc_summary.json"
import csv, glob, json, sys
from collections import defaultdict
vals = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/pmc_*/**/*_counter_collection.csv", recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            if "qmpc_lane_kernel" in row["Kernel_Name"]:
                vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
med = {k: sorted(v)[len(v) // 2] for k, v in vals.items()}
print(json.dumps({"launches": {k: len(v) for k, v in vals.items()}, "median_per_launch": med}, indent=1))
PY
find "$out" -name "*.csv" -size +4M -delete
