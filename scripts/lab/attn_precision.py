"""Lab: op-level error of the dh-80 attention kernels (knob attn_ring = 1: four-wave register-staged kernel with the running
fp32 maximum; 3: the ring kernel with the first-tile bf16 reference) against an fp64 reference on the same bf16 operands."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from adaface_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
for seed in range(4):
    g = torch.Generator().manual_seed(seed)
    B, N, heads, dh = 4, 1024, 8, 80
    sc = 1.0 + seed          # larger scores -> peakier softmax
    q = (torch.randn(B, N, heads * dh, generator=g) * sc ** 0.5).to(torch.bfloat16)
    k = (torch.randn(B, N, heads * dh, generator=g) * sc ** 0.5).to(torch.bfloat16)
    v = torch.randn(B, N, heads * dh, generator=g).to(torch.bfloat16)
    qd, kd, vd = (t.double().view(B, N, heads, dh).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qd @ kd.transpose(-1, -2) * dh ** -0.5, dim=-1) @ vd).transpose(1, 2).reshape(B, N, heads * dh)
    out = []
    for knob in (1, 3):
        _lib.set_knob("attn_ring", knob)
        got = ops.attention(q.to(dev), k.to(dev), v.to(dev), heads, dtype="bf16").double().cpu()
        e = got - ref
        out.append(f"ring={knob}: rms {e.pow(2).mean().sqrt().item():.3e} max {e.abs().max().item():.3e}")
    # the floor: the exact result rounded to bf16
    fl = ref.to(torch.bfloat16).double() - ref
    print(f"score scale {sc:.0f}: " + " | ".join(out) + f" | bf16 rounding of the exact result: rms {fl.pow(2).mean().sqrt().item():.3e} max {fl.abs().max().item():.3e}", flush=True)
