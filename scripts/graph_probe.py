#!/usr/bin/env python
"""Probe: does replaying the UNet forward as one hipGraph beat ~550 stream launches?  (SD-1.5, Bf = 16, bf16.)

    python scripts/graph_probe.py [--iters 20]
"""
import argparse
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--bf", type=int, default=16)
args = ap.parse_args()
dev = torch.device("cuda:0")
model = bench.build_model(dev, "bf16")
unet = model.model.diffusion_model
eng = unet.engine(dev)
from adaface_amd.synth import synth_context  # noqa: E402

B = args.bf
ctx = synth_context(B, seed=100, device=dev)
eng.set_context(ctx, B, True)
x = torch.randn(B, 4, 64, 64, device=dev)
t = torch.full((B,), 500, device=dev, dtype=torch.long)
out = torch.empty(B, 4, 64, 64, device=dev)
for _ in range(3):
    eng.unet_forward(x, t, out)
torch.cuda.synchronize()


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.iters * 1e3


eager = timed(lambda: eng.unet_forward(x, t, out))
ref = out.clone()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    eng.unet_forward(x, t, out)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    eng.unet_forward(x, t, out)
out.zero_()
g.replay()
torch.cuda.synchronize()
print("graph replay matches eager:", torch.equal(out, ref))
graph = timed(g.replay)
eager2 = timed(lambda: eng.unet_forward(x, t, out))
print(f"eager {eager:.3f} ms / forward, graph {graph:.3f} ms / forward, eager again {eager2:.3f} ms")
