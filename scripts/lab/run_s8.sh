mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_ops_gpu.py -x -q -k "conv2d_8x8" > gpurun_out/r4_s8.log 2>&1; echo "rc=$?" >> gpurun_out/r4_s8.log; tail -3 gpurun_out/r4_s8.log
grep -q "rc=0" gpurun_out/r4_s8.log || exit 1
python scripts/bench_shapes.py --only conv 2>&1 | grep "s1 up0" > gpurun_out/r4_s8_shapes.txt
echo "--- conv_halo8=7 (small-map kernel on 32x32 / 64x64 too)" >> gpurun_out/r4_s8_shapes.txt
python scripts/bench_shapes.py --only conv --knob conv_halo8=7 2>&1 | grep "s1 up0" >> gpurun_out/r4_s8_shapes.txt
cat gpurun_out/r4_s8_shapes.txt
