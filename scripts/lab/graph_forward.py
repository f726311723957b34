#!/usr/bin/env python
"""Lab: the twin forward (Bf = 16) launched eagerly vs replayed from a captured HIP graph (torch.cuda.CUDAGraph on a side stream)."""
import argparse, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from adaface_amd import _lib  # noqa: E402
from adaface_amd.engine import Engine  # noqa: E402
from adaface_amd.synth import synth_weights_into  # noqa: E402
from oracle import ldm_oracle as O  # noqa: E402  (parameter shapes only)
ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = O.SD15_UNET
kw = dict(in_channels=cfg.in_channels, model_channels=cfg.model_channels, out_channels=cfg.out_channels,
          num_res_blocks=cfg.num_res_blocks, attention_resolutions=cfg.attention_resolutions, channel_mult=cfg.channel_mult,
          num_heads=cfg.num_heads, context_dim=cfg.context_dim, transformer_depth=cfg.transformer_depth,
          n_context_layers=cfg.n_context_layers)
g = torch.Generator().manual_seed(3)
eng = Engine(dtype="bf16", unet=kw)
synth_weights_into(eng, O.unet_param_shapes(cfg), seed=1, device=dev)
x = torch.randn(8, 4, 64, 64, generator=g).to(dev)
t = torch.full((8,), 500, dtype=torch.long, device=dev)
ctx = torch.randn(16 * 16, 77, 768, generator=g).to(dev)
eng.set_context(ctx, 16, layerwise=True)
out = torch.empty(16, 4, 64, 64, device=dev)
def timeit(fn):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(args.reps): fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / args.reps)
    return best * 1e3
eager = lambda: eng.unet_forward_twin(x, t, out)
for _ in range(3): eager()
torch.cuda.synchronize()
ref = out.clone()
print(f"eager: {timeit(eager):.3f} ms", flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): eng.unet_forward_twin(x, t, out)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, stream=s):
    eng.unet_forward_twin(x, t, out)
out.zero_()
gr.replay(); torch.cuda.synchronize()
print("graph replay equals eager:", torch.equal(out, ref), flush=True)
print(f"graph: {timeit(gr.replay):.3f} ms", flush=True)
print(f"eager again: {timeit(eager):.3f} ms", flush=True)
