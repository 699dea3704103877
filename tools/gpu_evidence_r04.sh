#!/bin/bash
# Round-4 evidence on the GPU box, all on the SAME build: GPU tests, phase cycles, the default bench line, rocprofv3 kernel
# statistics of the default command, counter passes (tools/pmc_r04.sh) of the contract workload, a mid-size batch and
# the two large-batch legs.  usage (through gpurun): bash tools/gpu_evidence_r04.sh [tag]   -> gpurun_out/<tag>/
set -u
tag=${1:-r04_evidence}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
git -C "$root" rev-parse HEAD > "$out/commit.txt" 2>/dev/null || true
sha256sum "$root"/quaternion-mpc_amd/csrc/libqmpc_hip.so > "$out/lib_sha256.txt"
timeout 900 python -m pytest tests -m gpu -q > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
timeout 300 python tools/phase_profile.py > "$out/phase_cycles.txt" 2>&1
QMPC_WFORM=0 timeout 300 python tools/phase_profile.py > "$out/phase_cycles_round1_kernel.txt" 2>&1
timeout 900 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
timeout 300 python tools/wform_check.py --reps 30 > "$out/wform_vs_round1_b1024.txt" 2>&1
for b in 2048 4096 8192 16384; do timeout 300 python tools/wform_check.py --batch $b --oracle 64 --reps 8 2>&1 | grep -E "WFORM|wform vs" >> "$out/wform_vs_round1_midsize.txt"; done
timeout 600 python tools/handoff_bench.py --cases 10:32768,10:65536,20:65536,10:262144 --caps 0,default > "$out/handoff.txt" 2>&1
timeout 900 python tools/refmode_bench.py > "$out/refmode_bench.txt" 2>&1
timeout 900 python tools/refmode_lane_bench.py --cases 10:16384,10:32768,10:40960,10:65536,10:262144,20:16384,20:32768,20:65536 > "$out/refmode_lane.txt" 2>&1
for b in 1024 4096 16384 65536; do for f in 0 1; do QMPC_LOOP_FUSED=$f timeout 300 python tools/loop_bench.py --robots $b --mode 1 --ticks 60 2>&1 | tail -1 >> "$out/loop_refmode.txt"; done; done
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d "$out/prof" -- python "$root/bench.py" > "$out/prof_bench.json" 2> "$out/prof_bench.err"
cd "$root"
bash tools/pmc_r04.sh $tag/pmc_b1024_n10 1024 10 > /dev/null 2>&1
bash tools/pmc_r04.sh $tag/pmc_b8192_n10 8192 10 > /dev/null 2>&1
bash tools/pmc_r04.sh $tag/pmc_b32768_n10 32768 10 > /dev/null 2>&1
bash tools/pmc_r04.sh $tag/pmc_b65536_n20 65536 20 > /dev/null 2>&1
find "$out" -name "*.csv" -size +6M -delete
find "$out" -name "*_agent_info.csv" -delete
ls "$out"
tail -3 "$out/pytest_gpu.log"; head -c 400 "$out/bench_default.json"
