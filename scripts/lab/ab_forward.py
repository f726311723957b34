#!/usr/bin/env python
"""Lab: time the full SD-1.5 UNet forward at Bf = 16 (bf16, synthetic weights) with a given build of libadaface_hip.so,
so that two builds (e.g. the round-1 library and HEAD) can be compared on ONE box:
    python scripts/lab/ab_forward.py [--lib path/to/lib.so] [--reps 20] [--fp8]
Older builds lack newer entry points: missing symbols are skipped when --lib is given (lab only)."""
import argparse
import ctypes as C
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from adaface_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default="")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--fp8", action="store_true")
ap.add_argument("--knob", action="append", default=[])
ap.add_argument("--twin", action="store_true", help="af_unet_forward_twin on x[:8] (the CFG batch [x; x]) instead of af_unet_forward")
args = ap.parse_args()
if args.lib:
    _lib._LIB_PATH = Path(args.lib).resolve()
    probe = C.CDLL(str(_lib._LIB_PATH))
    _lib._SIGS[:] = [s for s in _lib._SIGS if hasattr(probe, s[0])]
from adaface_amd.engine import Engine  # noqa: E402
from adaface_amd.synth import synth_weights_into  # noqa: E402
from oracle import ldm_oracle as O  # noqa: E402  (parameter shapes only)

for kv in args.knob:
    k_, v_ = kv.split("=")
    _lib.set_knob(k_, int(v_))
dev = torch.device("cuda:0")
cfg = O.SD15_UNET
kw = dict(in_channels=cfg.in_channels, model_channels=cfg.model_channels, out_channels=cfg.out_channels,
          num_res_blocks=cfg.num_res_blocks, attention_resolutions=cfg.attention_resolutions, channel_mult=cfg.channel_mult,
          num_heads=cfg.num_heads, context_dim=cfg.context_dim, transformer_depth=cfg.transformer_depth,
          n_context_layers=cfg.n_context_layers)
eng = Engine(dtype="bf16", unet=kw)
synth_weights_into(eng, O.unet_param_shapes(cfg), seed=1, device=dev)
if args.fp8:
    eng.set_fp8(True)
g = torch.Generator().manual_seed(3)
x = torch.randn(16, 4, 64, 64, generator=g).to(dev)
t = torch.full((16,), 500, dtype=torch.long, device=dev)
ctx = torch.randn(16 * 16, 77, 768, generator=g).to(dev)
eng.set_context(ctx, 16, layerwise=True)
out = torch.empty_like(x)
if args.twin:
    xh, th = x[:8].contiguous(), t[:8].contiguous()
    fwd = lambda x_, t_, o_: eng.unet_forward_twin(xh, th, o_)
else:
    fwd = eng.unet_forward
for _ in range(3):
    fwd(x, t, out)
torch.cuda.synchronize()
best = 1e9
for rnd in range(3):
    t0 = time.perf_counter()
    for _ in range(args.reps):
        fwd(x, t, out)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / args.reps)
# per-class kernel time of one forward (HIP-event brackets around every launch: slows the forward, classes comparable)
lib = _lib.load()
lib.af_prof_reset(); lib.af_prof_set_stride(1) if hasattr(lib, "af_prof_set_stride") else None
lib.af_prof_enable(0x3ff)
fwd(x, t, out)
torch.cuda.synchronize()
lib.af_prof_enable(0)
n = 10
ms = (C.c_double * n)(); la = (C.c_int64 * n)(); fl = (C.c_double * n)(); by = (C.c_double * n)()
lib.af_prof_collect(n, ms, la, fl, by)
names = ["gemm_other", "attention", "groupnorm", "layernorm", "other", "pp160_gather", "pp160_plain", "pp128", "fp8", "halo8"]
print("   per-class ms per forward: " + ", ".join(f"{names[i]} {ms[i]:.2f} ({la[i]})" for i in range(n) if la[i])
      + f" | gemm total {ms[0] + ms[5] + ms[6] + ms[7] + ms[8] + ms[9]:.2f}")
print(f"{args.lib or 'HEAD'}{' fp8' if args.fp8 else ''}{' twin' if args.twin else ''} {' '.join(args.knob)}: UNet forward Bf=16: {best * 1e3:.3f} ms  (-> {8 / (50 * best + 0.026):.2f} images/s at 50 steps + 26 ms VAE)")
