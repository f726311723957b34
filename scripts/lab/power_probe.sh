#!/bin/bash
# Lab: sample power / clocks (rocm-smi) while the UNet forward runs back to back, and while one conv shape runs alone.
cd $GRAFT_REPO_ROOT
( for i in $(seq 1 24); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket|sclk|mclk|fclk|Temperature \(Sensor (edge|junction|hotspot)" | tr '\n' ';' ; echo; sleep 0.5; done ) > gpurun_out/power_samples.txt &
SP=$!
python scripts/lab/ab_forward.py --reps 150 2>&1 | grep -v amdgpu > gpurun_out/power_fwd.txt
wait $SP
cat gpurun_out/power_fwd.txt
cut -c1-400 gpurun_out/power_samples.txt
