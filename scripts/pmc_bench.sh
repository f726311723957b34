#!/bin/bash
# HBM traffic of the conv/linear class over one bench step: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), kernel
# trace only (no hip/hsa tracing beside counters), restricted to the class's kernels; post-processed by pmc_traffic.py
# into profiles/traffic_latest.json (read by bench.py for roofline.traffic).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
RE="conv_gemm_pp_kernel|conv_gemm_kernel|conv3x3_halo|splitk_reduce"   # (conv3x3_halo matches conv3x3_halo8_kernel too)
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w; mkdir -p gpurun_out/pmc_f gpurun_out/pmc_w
rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "$RE" --output-format csv -d gpurun_out/pmc_f -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-parity-leg --no-inflight-leg > gpurun_out/pmc_f/out.txt 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex "$RE" --output-format csv -d gpurun_out/pmc_w -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-parity-leg --no-inflight-leg > gpurun_out/pmc_w/out.txt 2>&1 || exit 1
python scripts/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/traffic.json > gpurun_out/traffic_print.txt 2>&1
find gpurun_out/pmc_f gpurun_out/pmc_w -name "*.csv" -size +1M -delete
