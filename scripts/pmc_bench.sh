#!/bin/bash
# HBM traffic of the conv/linear class over one bench step: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), kernel
# trace only (no hip/hsa tracing beside counters), restricted to the class's kernels; post-processed by pmc_traffic.py
# into profiles/traffic_latest.json (read by bench.py for roofline.traffic).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
RE="conv_gemm_pp_kernel|conv_gemm_kernel|conv3x3_halo|conv3x3_s8|splitk_reduce"   # (conv3x3_halo matches conv3x3_halo8_kernel too)
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w; mkdir -p gpurun_out/pmc_f gpurun_out/pmc_w
rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "$RE" --output-format csv -d gpurun_out/pmc_f -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-parity-leg > gpurun_out/pmc_f/out.txt 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex "$RE" --output-format csv -d gpurun_out/pmc_w -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-parity-leg > gpurun_out/pmc_w/out.txt 2>&1 || exit 1
# L2-miss reads by destination (exact 32-byte units, no x2 correction) and the L2 hit rate of the same kernels: the memory-side
# counters say "DRAM (MC)" for everything behind the L2 -- Infinity-Cache hits included; nothing rocprofv3 lists sees past it
for c in TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum; do
  rm -rf gpurun_out/pmc_$c; mkdir -p gpurun_out/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "conv3x3_halo8|conv3x3_s8" --output-format csv -d gpurun_out/pmc_$c -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-parity-leg > gpurun_out/pmc_$c/out.txt 2>&1 || exit 1
done
python scripts/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/traffic.json > gpurun_out/traffic_print.txt 2>&1
python - <<'PY' > gpurun_out/l2_side_counters.txt
import csv, glob, collections, re
out = collections.defaultdict(dict)
for c in ("TCC_EA0_RDREQ_DRAM_32B_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_HIT_sum", "TCC_MISS_sum"):
    tot = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                n = re.sub(r"\(.*", "", r["Kernel_Name"]); tot[n][0] += 1; tot[n][1] += float(r["Counter_Value"])
    for n, (k, v) in tot.items():
        out[n][c] = (k, v / max(k, 1))
for n, d in out.items():
    print(n, {c: (k, round(v, 1)) for c, (k, v) in d.items()})
    if "TCC_EA0_RDREQ_DRAM_32B_sum" in d: print("   L2-miss read bytes per launch (DRAM-destined, 32-B units):", round(d["TCC_EA0_RDREQ_DRAM_32B_sum"][1] * 32 / 1e6, 1), "MB")
    if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d: print("   L2 hit rate:", round(d["TCC_HIT_sum"][1] / (d["TCC_HIT_sum"][1] + d["TCC_MISS_sum"][1]), 3))
PY
for c in TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum; do find gpurun_out/pmc_$c -name "*.csv" -size +1M -delete; done
find gpurun_out/pmc_f gpurun_out/pmc_w -name "*.csv" -size +1M -delete
