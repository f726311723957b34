#!/bin/bash
# Whole-forward A/B of planner knobs on ONE device (scripts/lab/ab_forward.py per setting); run through scripts/gpu.sh.
#   bash scripts/lab/knob_sweep.sh "gemm_pp_minfill=30" "gemm_pp_minfill=70" ...
out=gpurun_out/knob_sweep.txt; : > $out
timeout -k 10 120 python scripts/lab/ab_forward.py --twin 2>/dev/null | tail -n 1 >> $out || exit 1
for kv in "$@"; do
  timeout -k 10 120 python scripts/lab/ab_forward.py --twin --knob $kv 2>/dev/null | tail -n 1 >> $out || exit 1
done
timeout -k 10 120 python scripts/lab/ab_forward.py --twin 2>/dev/null | tail -n 1 >> $out
cat $out
