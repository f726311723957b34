"""Lab: per-workgroup time line of conv3x3_halo8_kernel from in-kernel s_memrealtime stamps (a -DAF_LAB_ABLATE build:
scripts/lab/ablate_conv.sh build).  Shares, not lengths: the stamped build waits for its stores before the last stamp.

    python scripts/lab/halo_stamps.py [--cin 320 --cout 320 --hw 64]
"""
import argparse
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from adaface_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=320)
ap.add_argument("--cout", type=int, default=320)
ap.add_argument("--hw", type=int, default=64)
ap.add_argument("--lib", default="scripts/lab/ab/lib_lab.so")
args = ap.parse_args()
_lib._LIB_PATH = Path(args.lib).resolve()
lib = _lib.load()
from adaface_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.nn.functional.silu(torch.randn(16, args.cin, args.hw, args.hw, device=dev, generator=g))
w = torch.randn(args.cout, args.cin, 3, 3, device=dev, generator=g) * (args.cin * 9) ** -0.5
b = torch.randn(args.cout, device=dev, generator=g)
raw = C.CDLL(str(_lib._LIB_PATH))
raw.af_lab_stamps.argtypes = [C.c_void_p, C.c_int]
ntile = (16 * args.hw * args.hw // 256) * (args.cout // 160)


def bracketed(knob):
    _lib.set_knob("conv_fast_taps", knob)
    for _ in range(5):
        ops.conv2d(x, w, b, dtype="bf16")
    torch.cuda.synchronize()
    classes = (0, 5, 6, 7, 9)
    lib.af_prof_reset()
    lib.af_prof_enable(sum(1 << c for c in classes))
    for _ in range(20):
        ops.conv2d(x, w, b, dtype="bf16")
    torch.cuda.synchronize()
    lib.af_prof_enable(0)
    n = 10
    ms = (C.c_double * n)(); la = (C.c_int64 * n)(); fl = (C.c_double * n)(); by = (C.c_double * n)()
    lib.af_prof_collect(n, ms, la, fl, by)
    return sum(ms[c] for c in classes) / sum(la[c] for c in classes) * 1e3


print(f"event-bracketed launch: plain {bracketed(1):.1f} us, stamped build {bracketed(1 + 16 * 64):.1f} us")
for knob in (1 + 16 * 64,):
    _lib.set_knob("conv_fast_taps", knob)
    for _ in range(20):
        ops.conv2d(x, w, b, dtype="bf16")
    torch.cuda.synchronize()
    ops.conv2d(x, w, b, dtype="bf16")
    torch.cuda.synchronize()
    buf = np.zeros(2048 * 5, dtype=np.uint64)
    assert raw.af_lab_stamps(buf.ctypes.data, buf.size) == 0
    st = buf.reshape(2048, 5)[:ntile].astype(np.int64)
    t = st[:, :4] - st[:, 0].min()
    us = t / 100.0
    xcc = st[:, 4] >> 32
    hw = st[:, 4] & 0xFFFFFFFF
    cu = (hw >> 8) & 0xF
    se = (hw >> 13) & 0x7   # (field layout of HW_ID on gfx9: cu_id 11:8, sh_id 12, se_id 15:13)
    print(f"workgroups {ntile}; kernel span (first entry -> last store landed) {us[:, 3].max():.2f} us")
    order = np.argsort(us[:, 0])
    first = us[:, 0] < np.median(us[:, 0]) if ntile > 256 else np.ones(ntile, bool)
    for name, sel in (("first round", first), ("second round", ~first)):
        if sel.sum() == 0:
            continue
        u = us[sel]
        q = lambda a: "min %.2f / med %.2f / p90 %.2f / max %.2f" % (a.min(), np.median(a), np.percentile(a, 90), a.max())
        print(f"-- {name}: {sel.sum()} workgroups")
        print("   entry            ", q(u[:, 0]))
        print("   entry->loop      ", q(u[:, 1] - u[:, 0]))
        print("   loop             ", q(u[:, 2] - u[:, 1]))
        print("   epilogue+stores  ", q(u[:, 3] - u[:, 2]))
        print("   end              ", q(u[:, 3]))
    # per XCC: when its workgroups end
    for xc in sorted(set(xcc.tolist())):
        sel = xcc == xc
        print(f"   xcc {xc}: {sel.sum():4d} workgroups, entries {us[sel, 0].min():.2f}..{us[sel, 0].max():.2f}, last end {us[sel, 3].max():.2f}, mean loop {np.mean(us[sel, 2] - us[sel, 1]):.2f}")
