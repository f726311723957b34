mkdir -p gpurun_out
python scripts/bench_shapes.py --only xattn > gpurun_out/r4_xf_shapes.txt 2>&1
python scripts/bench_shapes.py --only attn >> gpurun_out/r4_xf_shapes.txt 2>&1
for i in 1 2; do
python scripts/lab/ab_forward.py --twin --knob xattn_fused=0 2>&1 | tail -2 >> gpurun_out/r4_xf_ab.txt
python scripts/lab/ab_forward.py --twin --knob xattn_fused=1 2>&1 | tail -2 >> gpurun_out/r4_xf_ab.txt
done
cat gpurun_out/r4_xf_shapes.txt gpurun_out/r4_xf_ab.txt
