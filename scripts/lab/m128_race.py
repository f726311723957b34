#!/usr/bin/env python
"""Lab: run the 128 x 160 tile GEMM repeatedly on one input and count launches whose output differs from the first / from
torch (race screen).  --stagger 77 = full vmcnt drain per step (debug)."""
import argparse, math, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from adaface_amd import _lib, ops
ap = argparse.ArgumentParser(); ap.add_argument("--stagger", type=int, default=-1); ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
_lib.load()
if a.stagger >= 0: _lib.set_knob("pp_stagger", a.stagger)
dev = torch.device("cuda:0")
for (M, K, N, res) in [(4096, 1280, 1280, False), (2048, 1280, 1280, False), (4096, 1280, 1280, True), (8192, 640, 640, False), (4096, 5120, 1280, False)]:
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).float().to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).float().to(dev)
    r = torch.randn(M, N, generator=g).to(torch.bfloat16).float().to(dev) if res else None
    ref = torch.nn.functional.linear(x, w) + (r if res else 0)
    first = None; bad_ref = 0; bad_first = 0; worst = 0.0
    for i in range(a.reps):
        y = ops.linear(x, w, None, r, dtype="bf16")
        e = (y - ref).abs().max().item() / ref.abs().max().item()
        worst = max(worst, e)
        bad_ref += e > 1.5e-2
        if first is None: first = y
        else: bad_first += (not torch.equal(y, first))
    print(f"[{M},{K}]->{N} res={res}: wrong vs torch {bad_ref}/{a.reps}, differs from first {bad_first}/{a.reps - 1}, worst rel err {worst:.3e}", flush=True)
