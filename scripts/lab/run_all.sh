mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r4_t3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t3.log; tail -3 gpurun_out/r4_t3.log
timeout -k 10 300 python scripts/lab/dbg_det.py 6 96 64 2>&1 | grep -v amdgpu.ids
