#!/bin/bash
# usage: scripts/pmc_shapes.sh <only> ; prints per-dispatch FETCH/WRITE (KiB) of conv_gemm kernels for bench_shapes --only <only> --reps 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ps_f gpurun_out/ps_w; mkdir -p gpurun_out/ps_f gpurun_out/ps_w
rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "conv_gemm_pp_kernel|conv_gemm_kernel|conv3x3_halo|attn_kernel" --output-format csv -d gpurun_out/ps_f -- python scripts/bench_shapes.py --only $1 --reps 1 > gpurun_out/ps_f/out.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex "conv_gemm_pp_kernel|conv_gemm_kernel|conv3x3_halo|attn_kernel" --output-format csv -d gpurun_out/ps_w -- python scripts/bench_shapes.py --only $1 --reps 1 > gpurun_out/ps_w/out.txt 2>&1
python - <<'PY'
import csv, glob
def rows(d, c):
    out=[]
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]==c: out.append((int(r["Dispatch_Id"]), r["Kernel_Name"][:60], r["Grid_Size"], float(r["Counter_Value"])))
    return sorted(out)
f=rows("gpurun_out/ps_f","FETCH_SIZE"); w=rows("gpurun_out/ps_w","WRITE_SIZE")
for a,b in zip(f,w):
    print(f"{a[1]:60s} grid={a[2]:>9s} fetch(x2)={2*a[3]/1024:9.1f} MiB write={b[3]/1024:9.1f} MiB")
PY
