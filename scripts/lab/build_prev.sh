#!/bin/bash
# Lab: build the library of the last COMMIT into scripts/lab/ab/lib_prev.so (git-ignored) for a same-device A/B against the
# working tree: bash scripts/lab/build_prev.sh && gpurun ... 'bash scripts/lab/ab_libs.sh scripts/lab/ab/lib_prev.so'
set -e
cd "$(dirname "$0")/../.."
rm -rf /tmp/af_prev && mkdir -p /tmp/af_prev scripts/lab/ab
git archive ${1:-HEAD} adaface_amd/csrc include | tar -x -C /tmp/af_prev
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form"
for f in /tmp/af_prev/adaface_amd/csrc/*.hip; do hipcc $F -c $f -o ${f%.hip}.o & done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/lab/ab/lib_prev.so /tmp/af_prev/adaface_amd/csrc/*.o
ls -la scripts/lab/ab/lib_prev.so
