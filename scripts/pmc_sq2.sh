#!/bin/bash
# usage: scripts/pmc_sq2.sh <only> "<counter list (<= 8 SQ counters)>"  -> the counters per dispatch, normalised by SQ_WAVE_CYCLES
# when it is in the list (first dispatch of every (kernel, grid) pair of bench_shapes --only <only> --reps 1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/sq2; mkdir -p gpurun_out/sq2
rocprofv3 --pmc $2 --kernel-trace --kernel-include-regex "conv_gemm_pp_kernel|conv3x3_halo8|rowpanel_kernel|xattn_short|attn_ring40|gemm_m128" --output-format csv -d gpurun_out/sq2 -- python scripts/bench_shapes.py --only $1 --reps 1 > gpurun_out/sq2/out.txt 2>&1
python - <<'PY'
import csv, glob, collections
rows=collections.OrderedDict()
for f in glob.glob("gpurun_out/sq2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=(int(r["Dispatch_Id"]), r["Kernel_Name"][:52], r["Grid_Size"])
        rows.setdefault(k,{})[r["Counter_Name"]]=float(r["Counter_Value"])
seen=set()
for k in sorted(rows):
    if (k[1],k[2]) in seen: continue
    seen.add((k[1],k[2]))
    v=rows[k]; wc=v.get("SQ_WAVE_CYCLES")
    print(f"{k[1]:52s} g={k[2]:>8s} " + " ".join(f"{n.replace('SQ_','')}={(x/wc if wc else x):.3g}" for n,x in sorted(v.items())))
PY
find gpurun_out/sq2 -name "*.csv" -size +200k -delete
