mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -k "cross_attention_layer or new_fused_paths or batch_consistency or layernorm_folding or forward_twin" > gpurun_out/r4_xf1.log 2>&1; echo "rc=$?" >> gpurun_out/r4_xf1.log; tail -5 gpurun_out/r4_xf1.log
grep -q "rc=0" gpurun_out/r4_xf1.log || exit 1
rm -f gpurun_out/r4_xf_ab.txt
for i in 1 2; do
python scripts/lab/ab_forward.py --twin --knob xattn_fused=0 2>&1 | tail -1 >> gpurun_out/r4_xf_ab.txt
python scripts/lab/ab_forward.py --twin --knob xattn_fused=1 2>&1 | tail -1 >> gpurun_out/r4_xf_ab.txt
done
cat gpurun_out/r4_xf_ab.txt
