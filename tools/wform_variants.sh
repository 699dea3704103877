#!/bin/bash
# link one libqmpc_hip.so per object quaternion-mpc_amd/csrc/build/wf_*.o (variants of qmpc_wform.hip compiled by hand)
# into tools/.prof/wf_*.so; on the GPU box: `tools/wform_variants.sh run` times each on the contract workload
B=quaternion-mpc_amd/csrc/build
mkdir -p tools/.prof
if [ "$1" = "run" ]; then
  for round in 1 2; do
  for v in tools/.prof/wf_*.so; do
    echo -n "$(basename $v .so): "
    QMPC_LIB=$PWD/$v python tools/wform_check.py --oracle 64 --reps 30 2>/dev/null | grep -E "WFORM=1|oracle on" | sed -e 's/.*kernel ms/ms/' -e 's/status ok.*//' | tr '\n' ' '
    echo
  done
  done
  exit 0
fi
for o in $B/wf_*.o; do
  n=$(basename $o .o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o tools/.prof/$n.so $B/qmpc_hip.o $B/qmpc_loop_fused.o $o $B/qmpc_lane.o $B/qmpc_lane_ref.o
done
ls tools/.prof/wf_*.so
