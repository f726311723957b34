mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -x -q -k "batch_consistency or other_latent_sizes or new_fused_paths or cross_attention_layer or conv2d_8x8" > gpurun_out/r4_s8.log 2>&1; echo "rc=$?" >> gpurun_out/r4_s8.log; tail -5 gpurun_out/r4_s8.log
grep -q "rc=0" gpurun_out/r4_s8.log || exit 1
rm -f gpurun_out/r4_s8_ab.txt
for i in 1 2; do
python scripts/lab/ab_forward.py --twin --knob conv_halo8=1 2>&1 | tail -1 >> gpurun_out/r4_s8_ab.txt
python scripts/lab/ab_forward.py --twin --knob conv_halo8=3 2>&1 | tail -1 >> gpurun_out/r4_s8_ab.txt
done
cat gpurun_out/r4_s8_ab.txt
python scripts/bench_shapes.py --only xattn 2>&1 | grep cross
