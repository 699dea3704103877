set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5a/pytest.log
tail -5 gpurun_out/r5a/pytest.log
timeout 600 python bench.py > gpurun_out/r5a/bench.json 2> gpurun_out/r5a/bench.err; tail -3 gpurun_out/r5a/bench.err
timeout 300 python tools/latency_b1.py 1000 > gpurun_out/r5a/latency.txt 2>&1
QMPC_WFORM_SMALL_LDS=0 timeout 300 python tools/latency_b1.py 300 > gpurun_out/r5a/latency_small0.txt 2>&1
cat gpurun_out/r5a/latency.txt
