#!/bin/bash
# gpurun_out/<tag>/ (tools/gpu_evidence_r05.sh) -> the round-5 files under profiles/ (summaries only; the raw counter CSVs stay
# in gpurun_out/).  usage: bash tools/collect_evidence_r05.sh <tag> <commit>
set -eu
tag=$1; commit=${2:-$(git rev-parse HEAD)}
src=gpurun_out/$tag
cp $src/bench_default.json profiles/r05_bench_default.json
cp $src/pytest_gpu.log profiles/r05_pytest_gpu.log
stats=$(find $src/prof -name "*kernel_stats.csv" | head -1)
cp "$stats" profiles/r05_rocprofv3_kernel_stats_default_cmd.csv
for f in phase_cycles latency_b1 refmode_check handoff wform_vs_round1_midsize loop_bench soak; do
  [ -f $src/$f.txt ] && cp $src/$f.txt profiles/r05_$f.txt
done
if [ -d $src/pmc_b1024_n10 ]; then
  cp $src/pmc_b1024_n10/r05_pmc_traffic*.json profiles/r05_pmc_traffic.json
  cp $src/pmc_b1024_n10/r05_sq_summary_*.json profiles/r05_sq_summary_b1024_n10.json
  for w in b8192_n10 b32768_n10 b65536_n20; do
    cp $src/pmc_$w/r05_pmc_traffic*.json profiles/r05_pmc_traffic_$w.json
    cp $src/pmc_$w/r05_sq_summary_*.json profiles/r05_sq_summary_$w.json
  done
fi
python tools/isa_metadata.py > profiles/r05_isa_metadata.txt 2>/dev/null
{
  echo "evidence of round 5 (profiles/r05_*): collected by tools/gpu_evidence_r05.sh in ONE gpurun call on the build of commit $commit"
  echo "libqmpc_hip.so sha256 on the GPU box: $(cut -d' ' -f1 $src/lib_sha256.txt)"
  echo "libqmpc_hip.so sha256 in the build container: $(sha256sum quaternion-mpc_amd/csrc/libqmpc_hip.so | cut -d' ' -f1)"
  echo "copied into profiles/ by tools/collect_evidence_r05.sh $tag"
} > profiles/r05_evidence_build.txt
cat profiles/r05_evidence_build.txt
