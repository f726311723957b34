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


def synth_context_adaprompt(batch: int, seed: int, device, n_layers: int = 16, n_tokens: int = 77, dim: int = 768,
                            subj_start: int = 6, n_subj_vectors: int = 16, subj_std: float = 0.07) -> torch.Tensor:
    """BASELINE.json configs[2] ("AdaPrompt embedding injected into the 77-token context") as synthetic data, SURVEY.md
    §8(d): a plain-prompt context whose 16 layer copies are identical per sample, with rows subj_start ..
    subj_start + 15 (the 16 subject vectors, --num_vectors_per_subj_token 16, stable_txt2img.py:258-260) of EVERY layer
    copy overwritten by per-layer draws of std ~0.07 (embedding_manager.py:1529) — so the 16 layer slices of a sample
    differ exactly where an AdaFace checkpoint would make them differ.  The real embeddings_gs-4500.pt is not in the
    tree; shapes, placement and scale follow the reference, the values are synthetic."""
    ctx = synth_context(batch, seed, "cpu", n_layers, n_tokens, dim).reshape(batch, n_layers, n_tokens, dim).clone()
    g = torch.Generator(device="cpu").manual_seed(seed + 7919)
    ctx[:, :, subj_start:subj_start + n_subj_vectors] = torch.randn(batch, n_layers, n_subj_vectors, dim, generator=g) * subj_std
    return ctx.reshape(batch * n_layers, n_tokens, dim).contiguous().to(device)


def synth_context_identity(batch: int, seed: int, device, n_layers: int = 16, n_tokens: int = 77, dim: int = 768,
                           id_start: int = 4, n_id_vectors: int = 16, id_dim: int = 512) -> torch.Tensor:
    """BASELINE.json configs[4] ("ArcFace identity-encoder context (arc2face path)") as synthetic data, SURVEY.md §8(d) /
    §8f-4: the zero-shot path turns a [1, 512] unit-norm ArcFace embedding into 16 identity token embeddings at text
    positions 4:20 (ldm/util.py:1085-1131, arc2face_models.py:175-280) through networks whose weights do not exist
    offline.  Here: a plain-prompt context whose rows id_start .. id_start + 15 of EVERY layer copy hold one seeded
    unit-norm 512-d vector per sample, zero-padded to 768 (util.py:1111) and scaled by sqrt(512) to the O(1) element scale
    of CLIP outputs — shape, placement and padding follow the reference, the projection itself is synthetic."""
    ctx = synth_context(batch, seed, "cpu", n_layers, n_tokens, dim).reshape(batch, n_layers, n_tokens, dim).clone()
    g = torch.Generator(device="cpu").manual_seed(seed + 104729)
    ident = torch.randn(batch, id_dim, generator=g)
    ident = ident / ident.norm(dim=1, keepdim=True) * math.sqrt(id_dim)
    row = torch.zeros(batch, dim)
    row[:, :id_dim] = ident
    ctx[:, :, id_start:id_start + n_id_vectors] = row[:, None, None, :]
    return ctx.reshape(batch * n_layers, n_tokens, dim).contiguous().to(device)
