#!/bin/bash
# One GPU-box pass that regenerates the judged profile artefacts at the current commit (run through scripts/gpu.sh):
#   AF_COMMIT=<sha> AF_TAG=r02 bash scripts/profile_round.sh
#   1. rocprofv3 --kernel-trace --stats of `bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing`
#      -> gpurun_out/${AF_TAG}_summary_${AF_COMMIT}.txt (+ rocprof's own kernel_stats csv)
#   2. the same in fp8 mode
#   3. HBM traffic of the conv/linear kernels: two separate --pmc passes (scripts/pmc_bench.sh) -> gpurun_out/traffic.json
# Copy what is to be judged from gpurun_out/ into profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${AF_TAG:-rXX}; SHA=${AF_COMMIT:-nocommit}
for mode in bf16 fp8; do
  D=gpurun_out/prof_$mode; rm -rf $D; mkdir -p $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python bench.py --dtype $mode --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-parity-leg --no-inflight-leg > $D/out.txt 2>&1 || exit 1
  T=$(find $D -name "*kernel_trace.csv" | head -1); S=$(find $D -name "*kernel_stats.csv" | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --dtype $mode --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-parity-leg --no-inflight-leg   (commit $SHA)"; tail -1 $D/out.txt | cut -c1-400; python scripts/summarize_trace.py $T; } > gpurun_out/${TAG}_summary_${mode}_${SHA}.txt
  cp $S gpurun_out/${TAG}_kernel_stats_${mode}_${SHA}.csv
  find $D -name "*.csv" -size +1M -delete
done
bash scripts/pmc_bench.sh || exit 1
cp gpurun_out/traffic.json gpurun_out/${TAG}_pmc_traffic_${SHA}.json
