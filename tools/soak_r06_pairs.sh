#!/bin/bash
# Round 6, after the pair forms' backward pass was split and the LDS staging rows went in: soak at batch sizes that leave the last
# wavefront partly filled and sit on the new switch-overs, both solver modes (tools/soak.py: every instance against the CPU oracle).
# usage through gpurun:  bash tools/soak_r06_pairs.sh > gpurun_out/soak_pairs.txt
s() { python tools/soak.py "$@" 2>&1 | grep -i soak; }
echo "== converged mode, lane pairs + hand-off: chunks of 14337 / 20011 / 32767 (N=10), 14849 / 30001 (N=20)"
s --instances 200718 --chunk 14337 --horizon 10; s --instances 300165 --chunk 20011 --horizon 10; s --instances 327670 --chunk 32767 --horizon 10
s --instances 103943 --chunk 14849 --horizon 20; s --instances 120004 --chunk 30001 --horizon 20
echo "== reference mode, lane pairs: chunks of 19457 / 30001 (N=10), 14849 / 17001 (N=20); full wavefronts 65535"
s --mode 1 --instances 194570 --chunk 19457 --horizon 10; s --mode 1 --instances 300010 --chunk 30001 --horizon 10
s --mode 1 --instances 103943 --chunk 14849 --horizon 20; s --mode 1 --instances 119007 --chunk 17001 --horizon 20; s --mode 1 --instances 131070 --chunk 65535 --horizon 10
