#!/bin/bash
# Round 6, review item 7: where do the persistent loop kernels with many spilled VGPRs (workspace forms, two waves per SIMD) still
# beat the per-tick form?  QMPC_LOOP_FUSED=1 / 0 on the configurations that select them.  usage (gpurun): bash tools/r06_loop_decide.sh
out=gpurun_out/r06_loop_decide.txt
: > $out
run() { for f in 1 0; do echo "$* fused $f: $(QMPC_LOOP_FUSED=$f timeout 300 python tools/loop_bench.py $* 2>&1 | tail -1 | cut -c1-260)" >> $out; done; }
run --robots 2048 --ticks 100                       # <5,.,false,false>: 41-65 spilled
run --robots 2048 --ticks 100 --warm 1 --mu0 1e-6
run --robots 4096 --ticks 100 --warm 1 --mu0 1e-6
run --robots 2048 --ticks 100 --mode 1              # <5,.,true,false>: 149-165 spilled
run --robots 1536 --ticks 100 --mode 1
run --robots 2048 --ticks 100 --horizon 20 --mode 1
run --robots 2048 --ticks 100 --model convex        # <1|2,.,false,true> / <5,..,true>: 99-124 spilled
run --robots 2048 --ticks 100 --model convex --horizon 20
run --robots 2048 --ticks 100 --model convex --horizon 20 --mode 1
run --robots 2048 --ticks 100 --horizon 20          # <6,...>
cat $out
