"""Base class of the HIP-backed drop-in modules.

A reference module (UNetModel, AutoencoderKL) is an nn.Module whose state_dict keys are
the weights contract.  The drop-in keeps that contract — parameters are
registered under exactly the reference's dotted names, so load_state_dict / .to / .eval /
state_dict work unchanged — while the arithmetic lives in an adaface_amd.engine.Engine
(repacked weights in HBM + HIP kernels).  The Engine is created on first use on a HIP
device and re-fed whenever the parameters may have changed.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Tuple

import torch
import torch.nn as nn


class _Node(nn.Module):
    """Bare container so that dotted reference names map onto nested attributes."""


def build_param_tree(root: nn.Module, shapes: Dict[str, Tuple[int, ...]], zero_init: Iterable[str] = ()) -> None:
    zero = set(zero_init)
    g = torch.Generator().manual_seed(0)
    for name, shape in shapes.items():
        parts = name.split(".")
        node = root
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, _Node())
            node = node._modules[p]
        if name in zero:
            t = torch.zeros(shape)
        elif len(shape) == 1:
            t = torch.ones(shape) if parts[-1] == "weight" else torch.zeros(shape)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            bound = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        node.register_parameter(parts[-1], nn.Parameter(t, requires_grad=False))


class HipModule(nn.Module):
    """nn.Module facade over an Engine.  Subclasses set `_engine_kwargs()` and `_ckpt_prefix`."""

    _ckpt_prefix = ""          # prefix the C library expects in front of this module's keys
    compute_dtype = "bf16"     # "bf16" (throughput), "f32" (parity mode) or "fp8" (bf16 + e4m3 ResBlock convolutions)

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_engine", None)
        object.__setattr__(self, "_weights_dirty", True)
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._mark_dirty())

    # ---- weight synchronisation -----------------------------------------------------------
    def _mark_dirty(self):
        object.__setattr__(self, "_weights_dirty", True)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._mark_dirty()
        return out

    def set_compute_dtype(self, dtype: str):
        """'bf16', 'f32' or 'fp8' (bf16 storage with the UNet's ResBlock 3x3 convolutions on the block-scaled fp8 MFMA;
        modules without such layers run it as bf16); takes effect at the next forward (the engine is rebuilt)."""
        if dtype not in ("bf16", "f32", "fp8"):
            raise ValueError(dtype)
        if dtype != self.compute_dtype:
            self.compute_dtype = dtype
            if self._engine is not None:
                self._engine.close()
            object.__setattr__(self, "_engine", None)
            self._mark_dirty()
        return self

    def _engine_kwargs(self) -> dict:  # pragma: no cover - abstract
        raise NotImplementedError

    def engine(self, device: torch.device):
        """The Engine on `device`, with current weights uploaded.  Raises if no HIP device."""
        from adaface_amd.engine import Engine
        if device.type != "cuda":
            raise RuntimeError(f"{type(self).__name__}: inputs must be on a HIP device (got {device}); "
                               "adaface_amd has no CPU path")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if self._engine is None or self._engine.device.index != idx:
            if self._engine is not None:
                self._engine.close()
            fp8 = self.compute_dtype == "fp8"
            object.__setattr__(self, "_engine", Engine(dtype="bf16" if fp8 else self.compute_dtype, device=idx,
                                                       **self._engine_kwargs()))
            if fp8:
                self._engine.set_fp8(True)
            self._mark_dirty()
        if self._weights_dirty:
            self.sync_weights()
        return self._engine

    def sync_weights(self):
        """Upload (repack) every parameter into the engine."""
        eng = self._engine
        if eng is None:
            return
        table = eng.tensor_table()
        sd = self.state_dict()
        for name in table:
            key = name[len(self._ckpt_prefix):] if name.startswith(self._ckpt_prefix) else name
            if key not in sd:
                raise KeyError(f"engine expects tensor '{name}' but the module has no parameter '{key}'")
            eng.load_tensor(name, sd[key])
        object.__setattr__(self, "_weights_dirty", False)
