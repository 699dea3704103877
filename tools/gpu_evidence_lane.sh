#!/bin/bash
# Round-3 evidence on the GPU box: kernel statistics of the default bench command (both kernel families appear: the
# contract launch and the large-batch legs) and the PMC passes of the lane kernel on the two large-batch configurations.
# usage (through gpurun): bash tools/gpu_evidence_lane.sh <tag>   -> gpurun_out/<tag>/
set -u
tag=${1:-evidence_lane}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d "$out/prof" -- python "$root/bench.py" --no-cpu-baseline --no-in-flight --no-closed-loop --no-reference-mode > "$out/prof_bench.json" 2> "$out/prof_bench.err"
cd "$root"
bash tools/pmc_lane.sh $tag/lane_b32768_n10 10:32768 > /dev/null 2>&1
bash tools/pmc_lane.sh $tag/lane_b65536_n20 20:65536 > /dev/null 2>&1
find "$out" -name "*.csv" -size +2M -delete
find "$out" -name "*_agent_info.csv" -delete
ls -R "$out" | head -40
