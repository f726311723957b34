#!/bin/bash
# rocprofv3 kernel trace of one bench step of a given workload (run through scripts/gpu.sh): bash scripts/lab/prof_config.sh config2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-config2}; D=gpurun_out/prof_$W; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-parity-leg > $D/out.txt 2>&1 || exit 1
T=$(find $D -name "*kernel_trace.csv" | head -1)
python scripts/summarize_trace.py $T > gpurun_out/prof_${W}_summary.txt
find $D -name "*.csv" -size +1M -delete
head -40 gpurun_out/prof_${W}_summary.txt | cut -c1-150
