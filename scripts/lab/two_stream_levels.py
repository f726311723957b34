#!/usr/bin/env python
"""Lab: would running the two CFG halves of ONE batch on two streams pay at the upper levels only?  Truncated UNets that
contain only the 64x64 level (channel_mult (1,)) or the 64x64 + 32x32 levels ((1, 2)): one Bf = 16 twin forward against two
Bf = 8 twin forwards side by side on two streams."""
import dataclasses, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from adaface_amd import _lib  # noqa: E402
from adaface_amd.engine import Engine  # noqa: E402
from adaface_amd.synth import synth_weights_into  # noqa: E402
from oracle import ldm_oracle as O  # noqa: E402  (parameter shapes only)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
def n_ctx_layers(cfg):
    inp, mid, out = O._unet_layout(cfg)
    return sum(1 for blk in inp + [mid] + out for l in blk if l[0] in ("xfmr", "attn", "st"))
def make(cfg, Bf, nl, hw=64):
    kw = dict(in_channels=cfg.in_channels, model_channels=cfg.model_channels, out_channels=cfg.out_channels,
              num_res_blocks=cfg.num_res_blocks, attention_resolutions=cfg.attention_resolutions, channel_mult=cfg.channel_mult,
              num_heads=cfg.num_heads, context_dim=cfg.context_dim, transformer_depth=cfg.transformer_depth, n_context_layers=nl)
    eng = Engine(dtype="bf16", unet=kw)
    synth_weights_into(eng, O.unet_param_shapes(cfg), seed=1, device=dev)
    x = torch.randn(Bf // 2, 4, hw, hw, generator=g).to(dev)
    t = torch.full((Bf // 2,), 500, dtype=torch.long, device=dev)
    ctx = torch.randn(nl * Bf, 77, 768, generator=g).to(dev)
    eng.set_context(ctx, Bf, layerwise=True)
    out = torch.empty(Bf, 4, hw, hw, device=dev)
    return eng, x, t, out
def run(engs, streams, reps=10):
    def once():
        for (eng, x, t, out), s in zip(engs, streams):
            with torch.cuda.stream(s):
                eng.unet_forward_twin(x, t, out)
    once(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps): once()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e3
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
# (round 4) the LOWER half on its own: a UNet of 1280 channels on a 16x16 latent = SD-1.5's 16x16 level (attention) + 8x8 level + middle block,
# whose kernels are launch- and latency-bound and fill half the chip or less at Bf = 8
for cm, ar, mc, hw in (((1, 1), (1,), 1280, 16), ((1,), (1,), 320, 64), ((1, 2), (2, 1), 320, 64), ((1, 2, 4, 4), (4, 2, 1), 320, 64)):
    cfg = dataclasses.replace(O.SD15_UNET, channel_mult=cm, attention_resolutions=ar, model_channels=mc)
    inp, mid, out = O._unet_layout(cfg)
    kinds = sorted({l[0] for blk in inp + [mid] + out for l in blk})
    nl = sum(1 for blk in inp + [mid] + out for l in blk if "xf" in l[0] or "attn" in l[0] or "transformer" in l[0])
    cfg = dataclasses.replace(cfg, n_context_layers=nl)
    print("model_channels", mc, "latent", hw, "channel_mult", cm, "layer kinds", kinds, "context layers", nl, flush=True)
    full = make(cfg, 16, nl, hw)
    one = run([full], [s0])
    del full
    a, b = make(cfg, 8, nl, hw), make(cfg, 8, nl, hw)
    half = run([a], [s0])
    two = run([a, b], [s0, s1])
    print(f"  one Bf=16 forward {one:.3f} ms | one Bf=8 {half:.3f} ms | two Bf=8 on two streams {two:.3f} ms ({two / one:.3f} of Bf=16)", flush=True)
    del a, b
