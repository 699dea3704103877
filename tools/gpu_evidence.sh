#!/bin/bash
# Collect the per-round evidence on the GPU box: GPU parity tests, phase cycles, bench lines for the
# BASELINE configurations, rocprofv3 kernel statistics and the two PMC traffic passes.
# usage (through gpurun): bash tools/gpu_evidence.sh <tag>      -> gpurun_out/<tag>/
set -u
tag=${1:-evidence}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
timeout 300 python tools/phase_profile.py > "$out/phase_cycles.txt" 2>&1
timeout 600 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
timeout 300 python bench.py --no-cpu-baseline --batch 32768 > "$out/bench_n10_b32768.json" 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --horizon 20 > "$out/bench_n20_b1024.json" 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --horizon 20 --batch 65536 --steps 10 --warmup 2 > "$out/bench_n20_b65536.json" 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --model convex --horizon 20 > "$out/bench_convex_n20_b1024.json" 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --model convex --horizon 20 --batch 32768 --steps 10 --warmup 2 > "$out/bench_convex_n20_b32768.json" 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --model biped8 --horizon 16 > "$out/bench_biped8_n16_b1024.json" 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --model biped8 --horizon 16 --batch 65536 --steps 3 --warmup 1 --check > "$out/bench_biped8_n16_b65536.json" 2>/dev/null
timeout 300 python tools/latency_b1.py > "$out/latency_b1.txt" 2>/dev/null
timeout 300 python tools/inflight_bench.py > "$out/inflight.txt" 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d "$out/prof" -- python "$root/bench.py" --no-cpu-baseline --no-in-flight --no-large-batch --no-closed-loop --no-reference-mode > "$out/prof_bench.json" 2> "$out/prof_bench.err"
timeout 600 rocprofv3 --pmc FETCH_SIZE -f csv -d "$out/pmc_fetch" -- python "$root/bench.py" --no-cpu-baseline --no-in-flight --no-large-batch --no-closed-loop --no-reference-mode --steps 10 > /dev/null 2> "$out/pmc_fetch.err"
timeout 600 rocprofv3 --pmc WRITE_SIZE -f csv -d "$out/pmc_write" -- python "$root/bench.py" --no-cpu-baseline --no-in-flight --no-large-batch --no-closed-loop --no-reference-mode --steps 10 > /dev/null 2> "$out/pmc_write.err"
cd "$root"
find "$out" -name "*.csv" -size +8M -delete
ls -R "$out" | head -50
tail -3 "$out/pytest_gpu.log"; cat "$out/bench_default.json"
