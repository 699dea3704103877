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
timeout 900 python -m pytest tests -m gpu -q > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
timeout 300 python bench.py --no-cpu-baseline --no-in-flight --no-closed-loop --no-reference-mode --batch 32768 --steps 10 --warmup 2 > "$out/bench_n10_b32768.json" 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-in-flight --no-closed-loop --no-reference-mode --horizon 20 --batch 65536 --steps 6 --warmup 2 > "$out/bench_n20_b65536.json" 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-in-flight --no-closed-loop --no-reference-mode --batch 65536 --steps 10 --warmup 2 > "$out/bench_n10_b65536.json" 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --no-in-flight --model biped8 --horizon 16 --batch 65536 --steps 3 --warmup 1 --check > "$out/bench_biped8_n16_b65536.json" 2>/dev/null
timeout 300 python tools/lane_bench.py --reps 3 --sample 64 --cases 10:16384,10:24576,10:32768,10:65536,10:262144,20:32768,20:65536 > "$out/lane_vs_wave.txt" 2>&1
timeout 300 python tools/lane_bench.py --model biped8 --reps 2 --sample 32 --cases 16:32768,16:65536 >> "$out/lane_vs_wave.txt" 2>&1
timeout 300 python tools/lane_bench.py --model convex --reps 2 --sample 32 --cases 20:16384,20:32768,20:65536,10:65536 >> "$out/lane_vs_wave.txt" 2>&1
# the closed loop at Monte-Carlo scale: lane kernel vs wave kernels (QMPC_LANE_MIN raised), cold and warm-started, both controllers
for r in 32768 65536; do
  for m in quat convex; do
    python tools/loop_bench.py --model $m --robots $r --ticks 60 2>&1 | tail -1
    python tools/loop_bench.py --model $m --robots $r --ticks 60 --warm 1 --mu0 1e-6 2>&1 | tail -1
    QMPC_LANE_MIN=100000000 python tools/loop_bench.py --model $m --robots $r --ticks 60 2>&1 | sed 's/^closed loop/[wave kernels] closed loop/' | tail -1
    QMPC_LANE_MIN=100000000 python tools/loop_bench.py --model $m --robots $r --ticks 60 --warm 1 --mu0 1e-6 2>&1 | sed 's/^closed loop/[wave kernels] closed loop/' | tail -1
  done
done > "$out/loop_large.txt" 2>&1
timeout 300 python bench.py --model convex --horizon 20 --batch 65536 --steps 4 --warmup 1 --no-cpu-baseline --no-in-flight --no-closed-loop --no-reference-mode > "$out/bench_convex_n20_b65536.json" 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1
(echo "== lane kernel soak, final build (chunk 32768 >= the switch-over: qmpc_lane_kernel)"; python tools/soak.py --instances 400000 --horizon 10 2>&1 | grep soak; python tools/soak.py --instances 100000 --horizon 20 2>&1 | grep soak; python tools/soak.py --model biped8 --instances 65536 --horizon 16 2>&1 | grep soak; python tools/soak.py --model convex --instances 131072 --horizon 20 2>&1 | grep soak
 echo "== closed loop on the lane kernel: 32768 robots x 300 ticks with random commands, 4 robots replayed on the host classes"; python tools/soak.py --closed-loop --robots 32768 --ticks 300 --feed-ang-vel 2>&1 | grep -v amdgpu | tail -1; python tools/soak.py --closed-loop --robots 32768 --ticks 300 --feed-ang-vel --warm 2>&1 | grep -v amdgpu | tail -1) > "$out/soak_lane.txt" 2>&1
tail -3 "$out/pytest_gpu.log"
