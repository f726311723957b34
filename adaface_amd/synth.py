"""Seeded synthetic weights / conditioning for benchmarking and smoke runs.

There is no network and no SD-1.5 checkpoint offline, so throughput is
measured on random-init weights of the exact SD-1.5 architecture and synthetic context of
the reference's shape ([B*16, 77, 768], the 16 layer copies identical per sample for a
plain-text prompt: embedding_manager.py:1349).  Generated directly in HBM.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch


def synth_weights_into(engine, shapes: Dict[str, Tuple[int, ...]], seed: int, device) -> None:
    """Draw every tensor of `shapes` (name -> shape) on the GPU and hand it to the engine."""
    g = torch.Generator(device=device).manual_seed(seed)
    for name in sorted(shapes):
        shp = tuple(shapes[name])
        if len(shp) == 1:
            t = torch.randn(shp, generator=g, device=device)
            t = 1.0 + 0.1 * t if name.endswith(".weight") else 0.05 * t
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            t = torch.randn(shp, generator=g, device=device) * (1.0 / math.sqrt(fan_in))
        engine.load_tensor(name, t)


def synth_context(batch: int, seed: int, device, n_layers: int = 16, n_tokens: int = 77, dim: int = 768,
                  shared: bool = False) -> torch.Tensor:
    """[batch*n_layers, n_tokens, dim] fp32 ~ N(0,1); the n_layers copies of a sample are identical.
    shared=True: one draw repeated over the batch (the unconditional prompt, stable_txt2img.py:630)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    n = 1 if shared else batch
    base = torch.randn(n, 1, n_tokens, dim, generator=g)
    if shared:
        base = base.expand(batch, 1, n_tokens, dim)
    ctx = base.expand(batch, n_layers, n_tokens, dim).reshape(batch * n_layers, n_tokens, dim).contiguous()
    return ctx.to(device)
