#!/usr/bin/env python
"""Lab: how much of a forward is launch-boundary idle time?  Two independent engines run the SAME twin forward (Bf = 16 or two
halves of Bf = 8) on two HIP streams at once; if two concurrent forwards take clearly less than twice one forward, workgroups of
one stream fill the drain / ramp bubbles of the other and a two-stream schedule of the CFG halves would pay."""
import argparse, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from adaface_amd import _lib  # noqa: E402
from adaface_amd.engine import Engine  # noqa: E402
from adaface_amd.synth import synth_weights_into  # noqa: E402
from oracle import ldm_oracle as O  # noqa: E402  (parameter shapes only)
ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=10); ap.add_argument("--knob", action="append", default=[])
args = ap.parse_args()
for kv in args.knob:
    k_, v_ = kv.split("="); _lib.set_knob(k_, int(v_))
dev = torch.device("cuda:0")
cfg = O.SD15_UNET
kw = dict(in_channels=cfg.in_channels, model_channels=cfg.model_channels, out_channels=cfg.out_channels,
          num_res_blocks=cfg.num_res_blocks, attention_resolutions=cfg.attention_resolutions, channel_mult=cfg.channel_mult,
          num_heads=cfg.num_heads, context_dim=cfg.context_dim, transformer_depth=cfg.transformer_depth,
          n_context_layers=cfg.n_context_layers)
g = torch.Generator().manual_seed(3)
def make(Bf):
    eng = Engine(dtype="bf16", unet=kw)
    synth_weights_into(eng, O.unet_param_shapes(cfg), seed=1, device=dev)
    x = torch.randn(Bf // 2, 4, 64, 64, generator=g).to(dev)
    t = torch.full((Bf // 2,), 500, dtype=torch.long, device=dev)
    ctx = torch.randn(16 * Bf, 77, 768, generator=g).to(dev)
    eng.set_context(ctx, Bf, layerwise=True)
    out = torch.empty(Bf, 4, 64, 64, device=dev)
    return eng, x, t, out
def run(engs, streams, reps):
    for (eng, x, t, out), s in zip(engs, streams):
        with torch.cuda.stream(s):
            eng.unet_forward_twin(x, t, out)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            for (eng, x, t, out), s in zip(engs, streams):
                with torch.cuda.stream(s):
                    eng.unet_forward_twin(x, t, out)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e3
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
for Bf in (16, 8):
    a, b = make(Bf), make(Bf)
    one = run([a], [s0], args.reps)
    seq = run([a, b], [s0, s0], args.reps)
    two = run([a, b], [s0, s1], args.reps)
    print(f"Bf={Bf}: one forward {one:.3f} ms | two forwards, one stream {seq:.3f} ms | two forwards, two streams {two:.3f} ms "
          f"({two / seq:.3f} of sequential)", flush=True)
    del a, b
