#!/bin/bash
# Round-6 evidence on the GPU box, all on the SAME build, one collection: GPU tests, the default bench line, rocprofv3 kernel
# statistics of the default command, hand-off / reference-mode / closed-loop tools, counter passes (tools/pmc_r06.sh) of the
# contract workload, a mid-size batch and the two large-batch legs, the soak.
# usage (through gpurun): bash tools/gpu_evidence_r06.sh [tag] [quick]   -> gpurun_out/<tag>/
set -u
tag=${1:-r06_evidence}
quick=${2:-}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
sha256sum "$root"/quaternion-mpc_amd/csrc/libqmpc_hip.so > "$out/lib_sha256.txt"
timeout 1500 python -m pytest tests -m gpu -q > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
timeout 900 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d "$out/prof" -- python "$root/bench.py" > "$out/prof_bench.json" 2> "$out/prof_bench.err"
cd "$root"
if [ -z "$quick" ]; then
  bash tools/pmc_r06.sh $tag/pmc_b1024_n10 1024 10 > /dev/null 2>&1
  bash tools/pmc_r06.sh $tag/pmc_b8192_n10 8192 10 > /dev/null 2>&1
  bash tools/pmc_r06.sh $tag/pmc_b32768_n10 32768 10 > /dev/null 2>&1
  bash tools/pmc_r06.sh $tag/pmc_b65536_n20 65536 20 > /dev/null 2>&1
  timeout 600 python tools/handoff_bench.py --cases 10:32768,10:65536,20:65536,10:262144 --caps 0,default > "$out/handoff.txt" 2>&1
  timeout 600 python tools/lane_switch_scan.py > "$out/lane_switch_scan.txt" 2>&1
  timeout 600 python tools/refmode_lane_bench.py --cases 10:32768,10:65536,20:65536 > "$out/refmode_lane.txt" 2>&1
  timeout 600 python tools/refmode_check.py 10 20 > "$out/refmode_check.txt" 2>&1
  timeout 300 python tools/latency_b1.py 1000 > "$out/latency_b1.txt" 2>&1
  for r in 32768 65536; do for w in 0 1; do
    if [ $w = 1 ]; then extra="--warm 1 --mu0 1e-6"; else extra=""; fi
    echo "robots $r warm $w: $(timeout 300 python tools/loop_bench.py --robots $r --ticks 60 $extra 2>&1 | tail -1)"
  done; done > "$out/loop_large.txt" 2>&1
  timeout 2400 bash tools/soak_r06.sh > "$out/soak.txt" 2>&1
fi
find "$out" -name "*.csv" -size +6M -delete
find "$out" -name "*_agent_info.csv" -delete
ls "$out"
tail -3 "$out/pytest_gpu.log"; head -c 600 "$out/bench_default.json"
