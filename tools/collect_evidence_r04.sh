#!/bin/bash
# gpurun_out/<tag>/ (tools/gpu_evidence_r04.sh) -> the round-4 files under profiles/ (summaries only; the raw counter CSVs stay
# in gpurun_out/).  usage: bash tools/collect_evidence_r04.sh <tag> <commit>
set -eu
tag=$1; commit=${2:-$(git rev-parse HEAD)}
src=gpurun_out/$tag
cp $src/bench_default.json profiles/r04_bench_default.json
cp $src/phase_cycles.txt profiles/r04_phase_cycles.txt
cp $src/phase_cycles_round1_kernel.txt profiles/r04_phase_cycles_round1_kernel.txt
cp $src/pytest_gpu.log profiles/r04_pytest_gpu.log
cp $src/wform_vs_round1_b1024.txt profiles/r04_wform_vs_round1_b1024.txt
cp $src/wform_vs_round1_midsize.txt profiles/r04_wform_vs_round1_midsize.txt
cp $src/handoff.txt profiles/r04_handoff.txt
cp $src/refmode_bench.txt profiles/r04_refmode_bench.txt
cp $src/refmode_lane.txt profiles/r04_refmode_lane.txt
cp $src/loop_refmode.txt profiles/r04_loop_refmode.txt
stats=$(find $src/prof -name "*kernel_stats.csv" | head -1)
cp "$stats" profiles/r04_rocprofv3_kernel_stats_default_cmd.csv
cp $src/pmc_b1024_n10/r04_pmc_traffic*.json profiles/r04_pmc_traffic.json
cp $src/pmc_b1024_n10/r04_sq_summary_*.json profiles/r04_sq_summary_b1024_n10.json
for w in b8192_n10 b32768_n10 b65536_n20; do
  cp $src/pmc_$w/r04_pmc_traffic*.json profiles/r04_pmc_traffic_$w.json
  cp $src/pmc_$w/r04_sq_summary_*.json profiles/r04_sq_summary_$w.json
done
python tools/isa_metadata.py > profiles/r04_isa_metadata.txt 2>/dev/null
{
  echo "evidence of round 4 (profiles/r04_*): collected by tools/gpu_evidence_r04.sh in ONE gpurun call on the build of commit $commit"
  echo "libqmpc_hip.so sha256 on the GPU box: $(cut -d' ' -f1 $src/lib_sha256.txt)"
  echo "libqmpc_hip.so sha256 in the build container: $(sha256sum quaternion-mpc_amd/csrc/libqmpc_hip.so | cut -d' ' -f1)"
  echo "copied into profiles/ by tools/collect_evidence_r04.sh $tag"
} > profiles/r04_evidence_build.txt
cat profiles/r04_evidence_build.txt
