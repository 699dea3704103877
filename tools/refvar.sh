#!/bin/bash
# reference-mode kernel time of every variant library tools/.prof/wf_*.so (GPU box)
for round in 1 2; do
for v in tools/.prof/wf_*.so; do
  echo -n "$(basename $v .so): "
  QMPC_LIB=$PWD/$v python tools/refmode_bench.py --cases 10:1024,10:8192 2>/dev/null | sed -e 's/.*wrench form/W/' -e 's/; status.*worst/ worst/' | tr '\n' ' '
  echo
done
done
