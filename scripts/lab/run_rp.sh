mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -k "linear or geglu or rowpanel or layernorm_fold or m128" > gpurun_out/r4_rp.log 2>&1; echo "rc=$?" >> gpurun_out/r4_rp.log; tail -4 gpurun_out/r4_rp.log
grep -q "rc=0" gpurun_out/r4_rp.log || exit 1
python scripts/bench_shapes.py --only linear 2>&1 | grep -v amdgpu > gpurun_out/r4_rp_shapes.txt
cat gpurun_out/r4_rp_shapes.txt
