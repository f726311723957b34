#!/usr/bin/env python
"""Per-shape kernel microbenchmark at the SD-1.5 / Bf=16 shapes (kernel-only time from the library's HIP-event
profiler, so the layout conversions of the op-level API are excluded).  Prints TFLOP/s or GB/s per shape.

    python scripts/bench_shapes.py [--dtype bf16] [--reps 5] [--only conv|linear|attn|norm|conv8]
    (conv8: the ResBlock convolutions with e4m3 operands next to their bf16 form, same process)
"""
import argparse
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from adaface_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--only", default="")
ap.add_argument("--bf", type=int, default=16)
ap.add_argument("--knob", action="append", default=[], help="name=value (af_knob_set), repeatable: A/B a kernel variant")
ap.add_argument("--lib", default="", help="lab only: another build of libadaface_hip.so to time on the same box (A/B of a structural change)")
args = ap.parse_args()
if args.lib:
    _lib._LIB_PATH = Path(args.lib).resolve()
    _probe = C.CDLL(str(_lib._LIB_PATH))
    _lib._SIGS[:] = [s_ for s_ in _lib._SIGS if hasattr(_probe, s_[0])]   # older builds lack newer entry points
lib = _lib.load()
for kv in args.knob:
    k, v = kv.split("=")
    _lib.set_knob(k, int(v))
dev = torch.device("cuda:0")
Bf = args.bf


GEMM_CLASSES = (0, 5, 6, 7, 9)   # four-wave / halo kernels + the three ping-pong instantiations (include/adaface_hip.h)


def timed(cls, fn):
    fn()  # warm
    torch.cuda.synchronize()
    classes = GEMM_CLASSES if cls == 0 else (cls,)
    lib.af_prof_reset()
    lib.af_prof_enable(sum(1 << c for c in classes))
    for _ in range(args.reps):
        fn()
    torch.cuda.synchronize()
    lib.af_prof_enable(0)
    n = 10
    ms = (C.c_double * n)(); la = (C.c_int64 * n)(); fl = (C.c_double * n)(); by = (C.c_double * n)()
    lib.af_prof_collect(n, ms, la, fl, by)
    tms, tla = sum(ms[c] for c in classes), sum(la[c] for c in classes)
    return tms / tla, sum(fl[c] for c in classes) / tla, sum(by[c] for c in classes) / tla


def show(name, ms, flops, byts):
    tf = flops / (ms * 1e-3) / 1e12 if flops else 0
    gb = byts / (ms * 1e-3) / 1e9
    print(f"{name:58s} {ms*1e3:9.1f} us  {tf:8.1f} TF/s  {gb:8.1f} GB/s(alg)", flush=True)


g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)

if args.only in ("", "conv"):
    convs = [  # Cin, Cout, H, ks, stride, up, per-forward count
        (320, 320, 64, 3, 1, 0, 13), (640, 320, 64, 3, 1, 0, 2), (960, 320, 64, 3, 1, 0, 1), (640, 640, 32, 3, 1, 1, 1),
        (320, 640, 32, 3, 1, 0, 1), (640, 640, 32, 3, 1, 0, 11), (1280, 640, 32, 3, 1, 0, 1), (1920, 640, 32, 3, 1, 0, 1),
        (1280, 1280, 16, 3, 1, 1, 1), (640, 1280, 16, 3, 1, 0, 1), (1280, 1280, 16, 3, 1, 0, 11),
        (2560, 1280, 16, 3, 1, 0, 2), (1280, 1280, 8, 3, 1, 0, 12), (2560, 1280, 8, 3, 1, 0, 3),
        (320, 320, 64, 3, 2, 0, 1), (1280, 1280, 16, 3, 2, 0, 1), (320, 320, 64, 1, 1, 0, 10), (640, 640, 32, 1, 1, 0, 10),
        (1280, 1280, 16, 1, 1, 0, 10), (128, 128, 512, 3, 1, 0, 0), (256, 256, 256, 3, 1, 0, 0),
    ]
    tot = 0.0
    for cin, cout, H, ks, st, up, cnt in convs:
        b = Bf if H <= 64 else 1
        x = rn(b, cin, H, H); w = rn(cout, cin, ks, ks) * (cin * ks * ks) ** -0.5; bias = rn(cout)
        ms, fl, by = timed(0, lambda: ops.conv2d(x, w, bias, stride=st, upsample=bool(up), dtype=args.dtype))
        show(f"conv{ks}x{ks} {cin}->{cout}@{H} s{st} up{up} B{b} (x{cnt})", ms, fl, by)
        tot += ms * cnt
    print(f"  conv weighted total per forward: {tot:.2f} ms")

if args.only == "conv8":
    convs = [(320, 320, 64, 13), (640, 320, 64, 2), (960, 320, 64, 1), (320, 640, 32, 1), (640, 640, 32, 11), (1280, 640, 32, 1),
             (1920, 640, 32, 1), (640, 1280, 16, 1), (1280, 1280, 16, 11), (2560, 1280, 16, 2), (1280, 1280, 8, 12), (2560, 1280, 8, 3)]
    tot = tot8 = 0.0
    for cin, cout, H, cnt in convs:
        x = torch.nn.functional.silu(rn(Bf, cin, H, H)); w = rn(cout, cin, 3, 3) * (cin * 9) ** -0.5; bias = rn(cout)
        ms, fl, by = timed(0, lambda: ops.conv2d(x, w, bias, dtype="bf16"))
        ms8, fl8, by8 = timed(8, lambda: ops.conv2d_fp8(x, w, bias))
        show(f"conv3x3 {cin}->{cout}@{H} B{Bf} bf16 (x{cnt})", ms, fl, by)
        show(f"conv3x3 {cin}->{cout}@{H} B{Bf} e4m3 (x{cnt})  [{ms / ms8:.2f}x]", ms8, fl8, by8)
        tot += ms * cnt
        tot8 += ms8 * cnt
    print(f"  ResBlock conv weighted total per forward: bf16 {tot:.2f} ms, e4m3 {tot8:.2f} ms")

if args.only in ("", "linear"):
    lins = [  # M, K, N, geglu, count
        (Bf * 4096, 320, 320, 0, 10), (Bf * 4096, 320, 960, 0, 5), (Bf * 4096, 320, 2560, 1, 5), (Bf * 4096, 1280, 320, 0, 5),
        (Bf * 1024, 640, 640, 0, 10), (Bf * 1024, 640, 1920, 0, 5), (Bf * 1024, 640, 5120, 1, 5), (Bf * 1024, 2560, 640, 0, 5),
        (Bf * 256, 1280, 1280, 0, 10), (Bf * 256, 1280, 3840, 0, 5), (Bf * 256, 1280, 10240, 1, 5), (Bf * 256, 5120, 1280, 0, 5),
        (Bf * 64, 1280, 1280, 0, 2), (Bf * 77, 768, 2560, 0, 0), (Bf, 1280, 18880, 0, 1),
    ]
    tot = 0.0
    for M, K, N, geglu, cnt in lins:
        x = rn(M, K); w = rn(N, K) * K ** -0.5; bias = rn(N)
        ms, fl, by = timed(0, lambda: ops.linear(x, w, bias, geglu=bool(geglu), dtype=args.dtype))
        show(f"linear [{M},{K}]->{N}{' geglu' if geglu else ''} (x{cnt})", ms, fl, by)
        tot += ms * cnt
    print(f"  linear weighted total per forward: {tot:.2f} ms")

if args.only in ("", "attn"):
    attns = [(4096, 4096, 40, 5), (1024, 1024, 80, 5), (256, 256, 160, 5), (64, 64, 160, 1), (4096, 77, 40, 5),
             (1024, 77, 80, 5), (256, 77, 160, 5)]
    tot = 0.0
    for Nq, Nk, dh, cnt in attns:
        q = rn(Bf, Nq, 8 * dh); k = rn(Bf, Nk, 8 * dh); v = rn(Bf, Nk, 8 * dh)
        ms, fl, by = timed(1, lambda: ops.attention(q, k, v, 8, dtype=args.dtype))
        show(f"attention N{Nq} S{Nk} d{dh} (x{cnt})", ms, fl, by)
        tot += ms * cnt
    print(f"  attention weighted total per forward: {tot:.2f} ms")

if args.only in ("", "attn", "xattn") and args.dtype == "bf16":
    # the cross-attention LAYER of a 64x64-level BasicTransformerBlock as one kernel (to_q + attention + to_out + residual),
    # next to the three launches it replaces (timed above: linear [M,320]->320 twice + attention N4096 S77 d40)
    N, S, Cn = 4096, 77, 320
    x = rn(Bf, N, Cn); gamma = rn(Cn) * 0.2 + 1; beta = rn(Cn) * 0.2; wq = rn(Cn, Cn) * Cn ** -0.5; wo = rn(Cn, Cn) * Cn ** -0.5
    bo = rn(Cn) * 0.1; kv = rn(Bf, S, 2 * Cn)
    ms, fl, by = timed(1, lambda: ops.xattn_fused(x, gamma, beta, wq, kv, wo, bo))
    show(f"cross-attention layer fused N{N} S{S} C{Cn} (x5)", ms, fl, by)

if args.only in ("", "norm"):
    gns = [(320, 64, 13), (640, 64, 2), (960, 64, 1), (320, 32, 1), (640, 32, 11), (1280, 32, 1), (1920, 32, 1),
           (640, 16, 1), (1280, 16, 11), (2560, 16, 2), (1280, 8, 12), (2560, 8, 3)]
    tot = 0.0
    for Cn, H, cnt in gns:
        x = rn(Bf, Cn, H, H); w = rn(Cn); b = rn(Cn)
        ms, fl, by = timed(2, lambda: ops.group_norm(x, w, b, silu=True, dtype=args.dtype))
        show(f"groupnorm C{Cn}@{H} (x{cnt})", ms, fl, by)
        tot += ms * cnt
    print(f"  groupnorm weighted total per forward: {tot:.2f} ms")
    for rows, Cn, cnt in [(Bf * 4096, 320, 15), (Bf * 1024, 640, 15), (Bf * 256, 1280, 15)]:
        x = rn(rows, Cn); w = rn(Cn); b = rn(Cn)
        ms, fl, by = timed(3, lambda: ops.layer_norm(x, w, b, dtype=args.dtype))
        show(f"layernorm [{rows},{Cn}] (x{cnt})", ms, fl, by)
