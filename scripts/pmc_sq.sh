#!/bin/bash
# usage: scripts/pmc_sq.sh <only>  -> SQ counters per dispatch for the GEMM/attention kernels of bench_shapes --only <only> --reps 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/sq; mkdir -p gpurun_out/sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --kernel-include-regex "conv_gemm_pp_kernel|conv_gemm_kernel|conv3x3_halo|attn_kernel|rowpanel_kernel|xattn_short|attn_ring40|gemm_m128" --output-format csv -d gpurun_out/sq -- python scripts/bench_shapes.py --only $1 --reps 1 > gpurun_out/sq/out.txt 2>&1
python - <<'PY'
import csv, glob, collections
rows=collections.OrderedDict()
for f in glob.glob("gpurun_out/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=(int(r["Dispatch_Id"]), r["Kernel_Name"][:58], r["Grid_Size"])
        rows.setdefault(k,{})[r["Counter_Name"]]=float(r["Counter_Value"])
names=["SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_VALU_MFMA_BUSY_CYCLES","SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE"]
seen=set()
for k in sorted(rows):
    if (k[1],k[2]) in seen: continue
    seen.add((k[1],k[2]))
    v=rows[k]; wc=v.get("SQ_WAVE_CYCLES",1)
    print(f"{k[1]:58s} g={k[2]:>8s} wave_cyc={wc:.3g} wait={v.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst={v.get('SQ_WAIT_INST_ANY',0)/wc:.2f} active={v.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} mfma_busy/busy={v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(1,v.get('SQ_BUSY_CYCLES',1)):.2f} ldsconf/ldsact={v.get('SQ_LDS_BANK_CONFLICT',0)/max(1,v.get('SQ_LDS_IDX_ACTIVE',1)):.2f}")
PY
