#!/bin/bash
# MFMA-pipe utilisation per kernel and shape: SQ_VALU_MFMA_BUSY_CYCLES (cycles the matrix pipes were busy, summed over
# SIMDs) against duration x 1024 SIMDs, once at the 2.4 GHz peak clock and once at the clock GRBM_GUI_ACTIVE implies.
# usage: scripts/pmc_mfma.sh  -> gpurun_out/mfma_util.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/mf; mkdir -p gpurun_out/mf
RE="conv_gemm_pp_kernel|conv_gemm_kernel|conv3x3_halo8_kernel|rowpanel_kernel|attn_kernel|attn_ring40_kernel|xattn_short_kernel"
for k in conv linear attn; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --kernel-include-regex "$RE" --output-format csv -d gpurun_out/mf/$k -- python scripts/bench_shapes.py --only $k --reps 1 > gpurun_out/mf/$k.txt 2>&1 || exit 1
done
python - <<'PY' > gpurun_out/mfma_util.txt
import csv, glob, collections
print("kernel | grid | duration us | MFMA busy / (duration x 1024 SIMD x 2.4 GHz) | GRBM_GUI_ACTIVE-implied clock GHz | busy / (GUI_ACTIVE x 1024)")
for k in ("conv", "linear", "attn"):
    cnt = collections.OrderedDict(); dur = {}
    for f in glob.glob(f"gpurun_out/mf/{k}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            key = int(r["Dispatch_Id"])
            cnt.setdefault(key, {"name": r["Kernel_Name"][:52], "grid": r["Grid_Size"]})[r["Counter_Name"]] = float(r["Counter_Value"])
    for f in glob.glob(f"gpurun_out/mf/{k}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
    order = sorted(cnt)
    for i, d in enumerate(order):                 # bench_shapes launches every shape twice (warm-up + timed): keep the
        v = cnt[d]                                # last of each run of identical (kernel, grid) dispatches
        nxt = cnt[order[i + 1]] if i + 1 < len(order) else None
        if d not in dur or (nxt is not None and (nxt["name"], nxt["grid"]) == (v["name"], v["grid"])): continue
        busy, gui, us = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0), dur[d]
        gui_x = gui / 8.0                          # summed over the 8 XCDs
        clk = gui_x / (us * 1e3) if us else 0.0
        print(f"{v['name']:52s} | {v['grid']:>9s} | {us:8.1f} | {busy / (us * 1e-6 * 2.4e9 * 1024):6.3f} | {clk:5.2f} | {busy / max(1.0, gui_x * 1024):6.3f}")
PY
find gpurun_out/mf -name "*.csv" -size +200k -delete
