#!/bin/bash
# Lab: timing ablations of the eight-wave conv kernel (results WRONG by construction).  Builds a second library with
# -DAF_LAB_ABLATE into scripts/lab/ab/lib_lab.so (git-ignored) and times the 3x3 convolutions with one phase removed.
#   scripts/lab/ablate_conv.sh build     (here: hipcc cross-compiles)
#   scripts/lab/ablate_conv.sh run       (on the GPU box)
set -e
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  mkdir -p scripts/lab/ab /tmp/af_lab
  F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form"
  for f in af_conv_gemm af_attention af_xattn_fused af_conv_s8 af_elementwise af_model af_norm af_ops; do
    hipcc $F $([ $f = af_conv_gemm ] && echo -DAF_LAB_ABLATE=1) -c adaface_amd/csrc/$f.hip -o /tmp/af_lab/$f.o &
  done
  wait
  hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/lab/ab/lib_lab.so /tmp/af_lab/*.o
  exit 0
fi
if [ "$1" = halo ]; then
  for k in 1 17 33 65 129 257 385 897 113; do
    echo "== halo kernel, conv_fast_taps=$k (+16 no LDS-DMA in the loop, +32 no reads, +64 no MFMAs, +128 no epilogue, +256 one K step, +512 no prologue staging)"
    python scripts/bench_shapes.py --only conv8 --lib scripts/lab/ab/lib_lab.so --knob conv_fast_taps=$k 2>&1 | grep "320->320@64 B16 bf16\|640->640@32 B16 bf16\|640->320@64 B16 bf16"
  done
  exit 0
fi
if [ "$1" = geglu ]; then
  for k in 1 17 33 65 129 97 225; do
    echo "== geglu row-panel, conv_fast_taps=$k (+16 no LDS-DMA, +32 no reads, +64 no MFMAs, +128 no epilogue)"
    python scripts/bench_shapes.py --only linear --lib scripts/lab/ab/lib_lab.so --knob conv_fast_taps=$k 2>&1 | grep "320\]->2560"
  done
  exit 0
fi
for k in 1 17 33 65 257 49 81 97; do
  echo "== conv_fast_taps=$k  (1 shipped; +16 no LDS-DMA in the loop; +32 no fragment reads; +64 no MFMAs; +256 activations of tap (0,0) only)"
  python scripts/bench_shapes.py --only conv8 --lib scripts/lab/ab/lib_lab.so --knob conv_fast_taps=$k 2>&1 | grep "320->320@64\|640->640@32\|1280->1280@16 "
done
