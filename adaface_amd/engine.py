"""Engine: Python owner of one af_handle (include/adaface_hip.h).

One Engine = the HIP-side twin of one reference nn.Module (a UNetModel or an
AutoencoderKL decoder): repacked weights + activation arena living in HBM, driven on the
current torch HIP stream.  PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, Optional, Sequence

import torch

from . import _lib
from ._lib import AfConfig, DTYPES, check, ptr, stream_ptr

UNET_PREFIX = "model.diffusion_model."
VAE_PREFIX = "first_stage_model."


def _fill(arr, values: Sequence[int]):
    if len(values) > 8:
        raise ValueError("at most 8 entries supported")
    for i, v in enumerate(values):
        arr[i] = int(v)


class Engine:
    def __init__(self, *, dtype: str = "bf16", device: int = 0, unet: Optional[dict] = None, vae: Optional[dict] = None,
                 clip: Optional[dict] = None):
        """unet: UNetModel ctor kwargs (in_channels, model_channels, out_channels, num_res_blocks,
        attention_resolutions, channel_mult, num_heads, context_dim, transformer_depth);
        vae: Decoder ddconfig (ch, out_ch, ch_mult, num_res_blocks, z_channels) + embed_dim."""
        self._lib = _lib.load()
        self._h = C.c_void_p()
        if not torch.cuda.is_available():
            raise _lib.AfError("adaface_amd.Engine needs a HIP device (there is no CPU path)")
        cfg = AfConfig()
        cfg.dtype = DTYPES[dtype]
        self.dtype = dtype
        self.device = torch.device("cuda", device)
        if unet is not None:
            cfg.build_unet = 1
            cfg.in_channels = unet["in_channels"]
            cfg.model_channels = unet["model_channels"]
            cfg.out_channels = unet["out_channels"]
            cfg.num_res_blocks = unet["num_res_blocks"]
            ar = list(unet["attention_resolutions"])
            cm = list(unet["channel_mult"])
            cfg.n_attention_resolutions = len(ar)
            _fill(cfg.attention_resolutions, ar)
            cfg.n_channel_mult = len(cm)
            _fill(cfg.channel_mult, cm)
            cfg.num_heads = unet["num_heads"]
            cfg.context_dim = unet["context_dim"]
            cfg.transformer_depth = unet.get("transformer_depth", 1)
            cfg.n_context_layers = unet.get("n_context_layers", 16)
        if vae is not None:
            cfg.build_vae = 1
            cfg.vae_ch = vae["ch"]
            cfg.vae_out_ch = vae["out_ch"]
            cfg.vae_num_res_blocks = vae["num_res_blocks"]
            cfg.vae_z_channels = vae["z_channels"]
            cfg.vae_embed_dim = vae.get("embed_dim", vae["z_channels"])
            vm = list(vae["ch_mult"])
            cfg.n_vae_ch_mult = len(vm)
            _fill(cfg.vae_ch_mult, vm)
            if vae.get("encoder", False):
                cfg.build_vae_encoder = 1
                cfg.vae_in_channels = vae.get("in_channels", 3)
        if clip is not None:   # CLIP text tower (vocab, hidden, layers, heads, intermediate, max_pos)
            cfg.build_clip = 1
            cfg.clip_vocab, cfg.clip_hidden, cfg.clip_layers = clip["vocab"], clip["hidden"], clip["layers"]
            cfg.clip_heads, cfg.clip_intermediate, cfg.clip_max_pos = clip["heads"], clip["intermediate"], clip["max_pos"]
        self.unet_cfg, self.vae_cfg, self.clip_cfg = unet, vae, clip
        torch.cuda.init()
        check(self._lib.af_create(device, C.byref(cfg), C.byref(self._h)), "af_create")
        self._ctx_key = None

    # -- lifetime -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.af_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights --------------------------------------------------------------------------
    def tensor_table(self) -> Dict[str, tuple]:
        """name -> expected shape, as the C library derived it."""
        n = self._lib.af_num_tensors(self._h)
        out = {}
        buf = (C.c_int64 * 4)()
        for i in range(n):
            name = self._lib.af_tensor_name(self._h, i).decode()
            nd = self._lib.af_tensor_shape(self._h, i, buf)
            out[name] = tuple(int(buf[d]) for d in range(nd))
        return out

    def missing_tensors(self):
        n = self._lib.af_num_tensors(self._h)
        return [self._lib.af_tensor_name(self._h, i).decode() for i in range(n)
                if not self._lib.af_tensor_loaded(self._h, i)]

    def load_tensor(self, name: str, t: torch.Tensor):
        t = t.detach().to(torch.float32).contiguous()
        shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
        fn = self._lib.af_load_tensor_device if t.is_cuda else self._lib.af_load_tensor
        check(fn(self._h, name.encode(), C.c_void_p(t.data_ptr()), t.dim(), shape), f"af_load_tensor({name})")

    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = "", strict: bool = True):
        """Feed every tensor the library expects from `sd` (keys = `prefix` + library name)."""
        table = self.tensor_table()
        for name in table:
            key = name if not prefix else name  # library names already carry the checkpoint prefix
            if key in sd:
                self.load_tensor(name, sd[key])
        missing = self.missing_tensors()
        if strict and missing:
            raise KeyError(f"{len(missing)} tensors missing from state_dict, e.g. {missing[:5]}")
        return missing

    # -- hot path -------------------------------------------------------------------------
    def set_context(self, ctx: torch.Tensor, Bf: int, layerwise: bool):
        """ctx: fp32 device tensor [Bf*16, T, D] (layerwise) or [Bf, T, D]."""
        if not ctx.is_cuda:
            raise ValueError("context must be a device tensor")
        ctx = ctx.contiguous().float()
        n_tokens = ctx.shape[1]
        L = self.unet_cfg.get("n_context_layers", 16) if layerwise else 1
        if ctx.shape[0] != Bf * L or ctx.shape[2] != self.unet_cfg["context_dim"]:
            raise ValueError(f"context shape {tuple(ctx.shape)} does not match batch {Bf} x layers {L}")
        check(self._lib.af_set_context(self._h, ptr(ctx), Bf, n_tokens, 1 if layerwise else 0, stream_ptr()),
              "af_set_context")

    def set_conv_attn(self, ks: int, batch_idx=(), token_idx=()):
        """Subject-token conv attention (attention.py:208-216), kernel size ks = 2, 3 or 4: sample `batch_idx[i]` carries a
        subject string whose first ks*ks token positions are `token_idx[i]` (a sample with several subject strings appears
        once per string).  ks <= 1 or no samples switches it off.  Invalidates the cached context."""
        n = len(batch_idx)
        bi = (C.c_int * max(n, 1))(*[int(b) for b in batch_idx])
        flat = [int(t) for row in token_idx for t in row]
        ti = (C.c_int * max(len(flat), 1))(*flat)
        if n and len(flat) != ks * ks * n:
            raise ValueError(f"set_conv_attn: {ks * ks} token positions per subject sample for a {ks}x{ks} kernel")
        check(self._lib.af_set_conv_attn(self._h, int(ks), n, bi, ti), "af_set_conv_attn")
        self._ctx_key = None

    def unet_forward(self, x: torch.Tensor, t: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = x.contiguous().float()
        t = t.contiguous().long()
        Bf, _, H, W = x.shape
        if out is None:
            out = torch.empty(Bf, self.unet_cfg["out_channels"], H, W, device=x.device, dtype=torch.float32)
        check(self._lib.af_unet_forward(self._h, ptr(x), ptr(t), ptr(out), Bf, H, W, stream_ptr()), "af_unet_forward")
        return out

    def unet_forward_twin(self, x: torch.Tensor, t: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """af_unet_forward_twin: the UNet on the CFG batch [x; x], [t; t] (context set for 2 * len(x) samples, cond first)
        without materialising the concatenation; the context-independent prefix runs once.  Returns eps [2B, C, H, W]."""
        x = x.contiguous().float()
        t = t.contiguous().long()
        B, _, H, W = x.shape
        if out is None:
            out = torch.empty(2 * B, self.unet_cfg["out_channels"], H, W, device=x.device, dtype=torch.float32)
        check(self._lib.af_unet_forward_twin(self._h, ptr(x), ptr(t), ptr(out), 2 * B, H, W, stream_ptr()),
              "af_unet_forward_twin")
        return out

    def unet_block_outputs(self, x: torch.Tensor, t: torch.Tensor, blocks=None) -> Dict[int, torch.Tensor]:
        """Outputs of U-Net blocks (forward order: input_blocks, middle_block, output_blocks) as fp32 NCHW tensors, one
        forward per requested block through the diagnostic tap (af_unet_set_tap).  Parity tests only."""
        x = x.contiguous().float()
        Bf, _, H, W = x.shape
        n = self._lib.af_unet_num_blocks(self._h)
        out = {}
        c, hh, ww = C.c_int(), C.c_int(), C.c_int()
        for b in (range(n) if blocks is None else blocks):
            check(self._lib.af_unet_block_shape(self._h, b, H, W, C.byref(c), C.byref(hh), C.byref(ww)), "af_unet_block_shape")
            buf = torch.empty(Bf, c.value, hh.value, ww.value, device=x.device, dtype=torch.float32)
            check(self._lib.af_unet_set_tap(self._h, b, ptr(buf)), "af_unet_set_tap")
            try:
                self.unet_forward(x, t)
            finally:
                self._lib.af_unet_set_tap(self._h, -1, None)
            out[b] = buf
        return out

    def vae_decode(self, z: torch.Tensor, scale_factor: float = 1.0, want_uint8: bool = False, want_float: bool = True):
        z = z.contiguous().float()
        B, _, H, W = z.shape
        f = 2 ** (len(self.vae_cfg["ch_mult"]) - 1)
        img = torch.empty(B, self.vae_cfg["out_ch"], H * f, W * f, device=z.device, dtype=torch.float32) if want_float else None
        u8 = torch.empty(B, H * f, W * f, 3, device=z.device, dtype=torch.uint8) if want_uint8 else None
        check(self._lib.af_vae_decode(self._h, ptr(z), float(scale_factor), ptr(img), ptr(u8), B, H, W, stream_ptr()),
              "af_vae_decode")
        if want_float and want_uint8:
            return img, u8
        return u8 if want_uint8 else img

    def vae_encode(self, x: torch.Tensor) -> torch.Tensor:
        """Encoder + quant_conv: x [B,3,H,W] -> posterior parameters [B, 2*embed_dim, H/f, W/f] (mean | logvar)."""
        x = x.contiguous().float()
        B, _, H, W = x.shape
        f = 2 ** (len(self.vae_cfg["ch_mult"]) - 1)
        ed = self.vae_cfg.get("embed_dim", self.vae_cfg["z_channels"])
        mom = torch.empty(B, 2 * ed, H // f, W // f, device=x.device, dtype=torch.float32)
        check(self._lib.af_vae_encode(self._h, ptr(x), ptr(mom), B, H, W, stream_ptr()), "af_vae_encode")
        return mom

    def posterior_sample(self, moments: torch.Tensor, noise: Optional[torch.Tensor], scale: float = 1.0) -> torch.Tensor:
        moments = moments.contiguous().float()
        B, C2, H, W = moments.shape
        z = torch.empty(B, C2 // 2, H, W, device=moments.device, dtype=torch.float32)
        nz = None if noise is None else noise.contiguous().float()
        check(self._lib.af_posterior_sample(ptr(moments), ptr(nz), float(scale), ptr(z), B, C2 // 2, H * W, stream_ptr()),
              "af_posterior_sample")
        return z

    # -- conditioning producer (CLIP text tower) --------------------------------------------
    def clip_embed_tokens(self, ids: torch.Tensor) -> torch.Tensor:
        """token_embedding(input_ids): int64 [B, T] device tensor -> fp32 [B, T, hidden]."""
        ids = ids.contiguous().long()
        out = torch.empty(*ids.shape, self.clip_cfg["hidden"], device=ids.device, dtype=torch.float32)
        check(self._lib.af_clip_embed_tokens(self._h, ptr(ids), ids.numel(), ptr(out), stream_ptr()), "af_clip_embed_tokens")
        return out

    def clip_text_forward(self, inputs_embeds: torch.Tensor, w_prev: float = 0.5, w_last: float = 0.5) -> torch.Tensor:
        """inputs_embeds fp32 [Bn, T, hidden] (before the position embeddings) -> blended, final-LayerNormed states."""
        x = inputs_embeds.contiguous().float()
        Bn, T, D = x.shape
        if D != self.clip_cfg["hidden"]:
            raise ValueError(f"inputs_embeds width {D} != hidden {self.clip_cfg['hidden']}")
        out = torch.empty_like(x)
        check(self._lib.af_clip_text_forward(self._h, ptr(x), Bn, T, float(w_prev), float(w_last), ptr(out), stream_ptr()),
              "af_clip_text_forward")
        return out

    def clip_text_forward3(self, inputs_embeds: torch.Tensor, w_prev2: float, w_prev: float, w_last: float) -> torch.Tensor:
        """As clip_text_forward with three blended hidden states (weights are used as given: normalise them first)."""
        x = inputs_embeds.contiguous().float()
        Bn, T, D = x.shape
        if D != self.clip_cfg["hidden"]:
            raise ValueError(f"inputs_embeds width {D} != hidden {self.clip_cfg['hidden']}")
        out = torch.empty_like(x)
        check(self._lib.af_clip_text_forward3(self._h, ptr(x), Bn, T, float(w_prev2), float(w_prev), float(w_last), ptr(out),
                                              stream_ptr()), "af_clip_text_forward3")
        return out

    def set_fp8(self, on: bool = True):
        """af_set_fp8: the UNet's ResBlock 3x3 convolutions read e4m3 activations / weights (bf16 engines only)."""
        check(self._lib.af_set_fp8(self._h, 1 if on else 0), "af_set_fp8")
        self.fp8 = bool(on)

    def arena_bytes(self) -> int:
        return int(self._lib.af_arena_bytes(self._h))
