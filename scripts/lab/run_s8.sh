mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_ops_gpu.py -x -q -k "conv2d or conv_gn" > gpurun_out/r4_s8.log 2>&1; echo "rc=$?" >> gpurun_out/r4_s8.log; tail -5 gpurun_out/r4_s8.log
grep -q "rc=0" gpurun_out/r4_s8.log || exit 1
python scripts/bench_shapes.py --only conv 2>&1 | grep "@16 s1 up0\|@8 " > gpurun_out/r4_s8_shapes.txt
python scripts/bench_shapes.py --only conv --knob conv_halo8=1 2>&1 | grep "@16 s1 up0\|@8 " >> gpurun_out/r4_s8_shapes.txt
cat gpurun_out/r4_s8_shapes.txt
rm -f gpurun_out/r4_s8_ab.txt
for i in 1 2; do
python scripts/lab/ab_forward.py --twin --knob conv_halo8=1 2>&1 | tail -1 >> gpurun_out/r4_s8_ab.txt
python scripts/lab/ab_forward.py --twin --knob conv_halo8=3 2>&1 | tail -1 >> gpurun_out/r4_s8_ab.txt
done
cat gpurun_out/r4_s8_ab.txt
