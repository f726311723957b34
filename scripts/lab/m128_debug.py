#!/usr/bin/env python
"""Lab: per-tile error map of the 128 x 160 tile GEMM (which rows / columns of a tile are wrong)."""
import math, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from adaface_amd import _lib, ops
_lib.load()
dev = torch.device("cuda:0")
import os
M, K, N = 4096, 1280, 1280
if os.environ.get("ST"): _lib.set_knob("pp_stagger", int(os.environ["ST"]))
g = torch.Generator().manual_seed(1)
x = torch.randn(M, K, generator=g).to(torch.bfloat16).float().to(dev)
w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).float().to(dev)
ref = torch.nn.functional.linear(x, w)
for rep in range(3):
    y = ops.linear(x, w, None, None, dtype="bf16")
    bad = ((y - ref).abs() > 0.05 * ref.abs().max())
    print("run", rep, "bad fraction", bad.float().mean().item(), "plan", _lib.plan_counts(reset=True)["rowpanel"])
    t = bad.view(M // 128, 128, N // 160, 160)
    per_tile = t.float().mean(dim=(1, 3))
    print(" tiles with errors:", int((per_tile > 0).sum()), "of", per_tile.numel(), " mean bad frac in bad tiles", per_tile[per_tile > 0].mean().item() if (per_tile > 0).any() else 0)
    rows = t.float().mean(dim=(0, 2, 3)); cols = t.float().mean(dim=(0, 1, 2))
    print(" bad by row-in-tile (16-row groups):", [round(rows[i*16:(i+1)*16].mean().item(), 3) for i in range(8)])
    print(" bad by col-in-tile (16-col groups):", [round(cols[i*16:(i+1)*16].mean().item(), 3) for i in range(10)])
    yb = y[bad]
    print(" bad values: zeros", (yb == 0).float().mean().item() if yb.numel() else None, " sample", yb[:6].tolist(), ref[bad][:6].tolist())
    tm, tn = [int(v[0]) for v in torch.nonzero(per_tile > 0, as_tuple=True)]
    sub = t[tm, :, tn, :]
    rc = torch.nonzero(sub).tolist()
    print(" first bad tile", (tm, tn), "bad (row, col) in tile:", rc[:40])
