#!/bin/bash
# Lab: timing ablations of xattn_fused_kernel (results WRONG by construction).  Builds a second library with -DAF_LAB_ABLATE
# into scripts/lab/ab/lib_lab.so (git-ignored) and times the fused cross-attention layer with one phase removed.
#   scripts/lab/ablate_xattn.sh build     (here: hipcc cross-compiles)
#   scripts/lab/ablate_xattn.sh run       (on the GPU box)
set -e
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  mkdir -p scripts/lab/ab /tmp/af_lab
  F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form"
  for f in af_conv_gemm af_attention af_xattn_fused af_conv_s8 af_elementwise af_model af_norm af_ops; do
    hipcc $F $([ $f = af_xattn_fused ] && echo -DAF_LAB_ABLATE=1) -c adaface_amd/csrc/$f.hip -o /tmp/af_lab/$f.o &
  done
  wait
  hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/lab/ab/lib_lab.so /tmp/af_lab/*.o
  exit 0
fi
for k in 1 17 33 65 129 257 $((1+16+32+64)) $((1+16+32+64+128+256)); do
  echo "== xattn_fused=$k  (1 shipped; +16 no phase-1 steps; +32 no attention; +64 no phase-3 steps; +128 no row loads; +256 no epilogue stores / residual)"
  python scripts/bench_shapes.py --only xattn --lib scripts/lab/ab/lib_lab.so --knob xattn_fused=$k 2>&1 | grep "cross-attention"
done
