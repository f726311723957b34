#!/bin/bash
# Whole-forward A/B of two builds on ONE device: bash scripts/lab/ab_libs.sh scripts/lab/ab/lib_prev.so [extra ab_forward args]
prev=$1; shift
out=gpurun_out/ab_libs.txt; : > $out
for r in 1 2; do
  timeout -k 10 120 python scripts/lab/ab_forward.py --twin --lib $prev "$@" 2>/dev/null | tail -n 1 >> $out || exit 1
  timeout -k 10 120 python scripts/lab/ab_forward.py --twin "$@" 2>/dev/null | tail -n 1 >> $out || exit 1
done
cat $out
