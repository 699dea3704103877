#!/bin/bash
# A/B of two builds of the library on the closed loop (same box): bash tools/loop_ab.sh <tag> <old.so> <new.so>
set -u
tag=$1; old=$2; new=$3
out=gpurun_out/$tag; mkdir -p $out
for rep in 1 2; do
for v in old new; do
  case $v in old) so=$old;; new) so=$new;; esac
  for cfg in "--robots 1024" "--robots 4096" "--robots 16384 --ticks 100" "--robots 1024 --warm 1" "--robots 1024 --mode 1" "--robots 4096 --model convex"; do
    echo "== $v rep $rep $cfg" >> $out/loop_ab.txt
    QMPC_LIB=$PWD/$so timeout 300 python tools/loop_bench.py $cfg 2>&1 | tail -2 >> $out/loop_ab.txt
  done
done
done
cat $out/loop_ab.txt
