#!/bin/bash
# Round-5 evidence on the GPU box, all on the SAME build, ONE collection per kernel build: GPU tests, the default bench
# line, rocprofv3 kernel statistics of the default command, phase cycles, hand-off / reference-mode / latency / closed-loop
# tools, counter passes (tools/pmc_r05.sh) of the contract workload, a mid-size batch and the two large-batch legs.
# usage (through gpurun): bash tools/gpu_evidence_r05.sh [tag] [quick]   -> gpurun_out/<tag>/
set -u
tag=${1:-r05_evidence}
quick=${2:-}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
sha256sum "$root"/quaternion-mpc_amd/csrc/libqmpc_hip.so > "$out/lib_sha256.txt"
timeout 1200 python -m pytest tests -m gpu -q > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
timeout 900 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d "$out/prof" -- python "$root/bench.py" > "$out/prof_bench.json" 2> "$out/prof_bench.err"
cd "$root"
if [ -z "$quick" ]; then
  timeout 300 python tools/phase_profile.py > "$out/phase_cycles.txt" 2>&1
  timeout 300 python tools/latency_b1.py 1000 > "$out/latency_b1.txt" 2>&1
  timeout 900 python tools/refmode_check.py 10 16 20 24 32 > "$out/refmode_check.txt" 2>&1
  timeout 600 python tools/handoff_bench.py --cases 10:32768,10:65536,20:65536,10:262144 --caps 0,default > "$out/handoff.txt" 2>&1
  for b in 2048 4096 8192 16384; do timeout 300 python tools/wform_check.py --batch $b --oracle 64 --reps 8 2>&1 | grep -E "WFORM|wform vs" >> "$out/wform_vs_round1_midsize.txt"; done
  for r in 1024 2048 4096; do for w in 0 1; do
    if [ $w = 1 ]; then extra="--warm 1 --mu0 1e-6"; else extra=""; fi
    for f in 1 0; do echo "robots $r warm $w fused $f: $(QMPC_LOOP_FUSED=$f timeout 300 python tools/loop_bench.py --robots $r --ticks 100 $extra 2>&1 | tail -1)"; done
  done; done > "$out/loop_bench.txt" 2>&1
  timeout 1500 bash tools/soak_r05.sh > "$out/soak.txt" 2>&1
  bash tools/pmc_r05.sh $tag/pmc_b1024_n10 1024 10 > /dev/null 2>&1
  bash tools/pmc_r05.sh $tag/pmc_b8192_n10 8192 10 > /dev/null 2>&1
  bash tools/pmc_r05.sh $tag/pmc_b32768_n10 32768 10 > /dev/null 2>&1
  bash tools/pmc_r05.sh $tag/pmc_b65536_n20 65536 20 > /dev/null 2>&1
fi
find "$out" -name "*.csv" -size +6M -delete
find "$out" -name "*_agent_info.csv" -delete
ls "$out"
tail -3 "$out/pytest_gpu.log"; head -c 400 "$out/bench_default.json"
