"""Build libadaface_hip.so (gfx950) in-tree with hipcc.

    python -m adaface_amd.build            # incremental
    python -m adaface_amd.build --force

The .so lands in adaface_amd/ (git-ignored, travels to the GPU box with the tree).
hipcc cross-compiles for gfx950 without a GPU present.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
BUILD = PKG / "_build"
LIB = PKG / "libadaface_hip.so"
SOURCES = ["af_conv_gemm.hip", "af_norm.hip", "af_attention.hip", "af_xattn_fused.hip", "af_conv_s8.hip", "af_elementwise.hip", "af_model.hip", "af_ops.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result", "-Wno-unused-value",
         # MFMA accumulators in VGPRs (gfx950 has one unified register file): no v_accvgpr_read/write around the
         # softmax / epilogue VALU work, and fewer registers in total
         "-mllvm", "-amdgpu-mfma-vgpr-form"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; cannot build libadaface_hip.so")
    return exe


def _newest_header() -> float:
    hs = list(CSRC.glob("*.h")) + [PKG.parent / "include" / "adaface_hip.h"]
    return max(h.stat().st_mtime for h in hs)


def _compile(src: str, force: bool) -> Path:
    obj = BUILD / (src + ".o")
    s = CSRC / src
    if not force and obj.exists() and obj.stat().st_mtime > max(s.stat().st_mtime, _newest_header()):
        return obj
    cmd = [_hipcc(), *FLAGS, "-c", str(s), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = True) -> Path:
    BUILD.mkdir(exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    if force or not LIB.exists() or any(o.stat().st_mtime > LIB.stat().st_mtime for o in objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[adaface_amd.build] {LIB} ({LIB.stat().st_size / 1e6:.1f} MB)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
