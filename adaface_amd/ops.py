"""Operator-level wrappers over the C ABI (af_op_* in include/adaface_hip.h).

Same names / argument meaning as the torch.nn.functional calls the reference makes on
the hot path, operating on fp32 CUDA(HIP) tensors in the reference's
layouts.  `dtype` selects the kernels' storage/MFMA type: "bf16" or "f32".
Used by the parity tests and by nobody on the hot path (the model executors call the
same kernels internally with no conversions).
"""
from __future__ import annotations

import math

import torch

from . import _lib
from ._lib import DTYPES, check, ptr, stream_ptr


def _dev_f32(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError("adaface_amd.ops: tensors must live on the GPU (no CPU path)")
    return t.contiguous().float()


def conv2d(x, weight, bias=None, stride=1, padding=None, upsample=False, residual=None, dtype="bf16"):
    """F.conv2d(F.interpolate(x, 2, 'nearest') if upsample else x, weight, bias, stride, padding) [+ residual]."""
    lib = _lib.load()
    x, weight = _dev_f32(x), _dev_f32(weight)
    B, Cin, H, W = x.shape
    Cout, Cin2, ks, ks2 = weight.shape
    if Cin2 != Cin or ks != ks2:
        raise ValueError(f"conv2d: weight {tuple(weight.shape)} does not match input {tuple(x.shape)}")
    pad = ks // 2 if padding is None else padding
    up = 1 if upsample else 0
    Hi, Wi = H << up, W << up
    Ho, Wo = (Hi + 2 * pad - ks) // stride + 1, (Wi + 2 * pad - ks) // stride + 1
    y = torch.empty(B, Cout, Ho, Wo, device=x.device, dtype=torch.float32)
    b = None if bias is None else _dev_f32(bias)
    r = None if residual is None else _dev_f32(residual)
    check(lib.af_op_conv2d(DTYPES[dtype], ptr(x), ptr(weight), ptr(b), ptr(r), ptr(y), B, Cin, H, W, Cout, ks, stride,
                           pad, up, stream_ptr()), "af_op_conv2d")
    return y


FP8_ACT_SHIFT = 3   # fp8 activations hold value * 2^3 (AF_FP8_ACT_SHIFT in csrc/af_model.hip)


def conv2d_fp8(x, weight, bias=None, stride=1, upsample=False, residual=None, act_shift=FP8_ACT_SHIFT):
    """conv2d with both operands quantised to OCP e4m3 as the UNet's fp8 mode does (x * 2^act_shift saturating, weight
    rows scaled by a power of two), bf16 bias / residual / output.  Raises AfError when the shape has no fp8 plan."""
    lib = _lib.load()
    x, weight = _dev_f32(x), _dev_f32(weight)
    B, Cin, H, W = x.shape
    Cout, Cin2, ks, ks2 = weight.shape
    if Cin2 != Cin or ks != ks2:
        raise ValueError(f"conv2d_fp8: weight {tuple(weight.shape)} does not match input {tuple(x.shape)}")
    pad, up = ks // 2, 1 if upsample else 0
    Hi, Wi = H << up, W << up
    Ho, Wo = (Hi + 2 * pad - ks) // stride + 1, (Wi + 2 * pad - ks) // stride + 1
    y = torch.empty(B, Cout, Ho, Wo, device=x.device, dtype=torch.float32)
    b = None if bias is None else _dev_f32(bias)
    r = None if residual is None else _dev_f32(residual)
    check(lib.af_op_conv2d_fp8(ptr(x), ptr(weight), ptr(b), ptr(r), ptr(y), B, Cin, H, W, Cout, ks, stride, pad, up,
                               act_shift, stream_ptr()), "af_op_conv2d_fp8")
    return y


def group_norm_fp8(x, weight, bias, eps=1e-5, silu=False, act_shift=FP8_ACT_SHIFT):
    """GroupNorm(32) [+ SiLU] written as e4m3 bytes of result * 2^act_shift: uint8 [B, H*W, C] (NHWC)."""
    lib = _lib.load()
    x = _dev_f32(x)
    B, Cn, H, W = x.shape
    y = torch.empty(B, H * W, Cn, device=x.device, dtype=torch.uint8)
    check(lib.af_op_groupnorm_fp8(ptr(x), ptr(_dev_f32(weight)), ptr(_dev_f32(bias)), eps, 1 if silu else 0, ptr(y), B, Cn,
                                  H, W, act_shift, stream_ptr()), "af_op_groupnorm_fp8")
    return y


def layer_norm_fp8(x, weight, bias, eps=1e-5, act_shift=FP8_ACT_SHIFT):
    """LayerNorm over the last dim written as e4m3 bytes of result * 2^act_shift: uint8, same shape as x."""
    lib = _lib.load()
    x = _dev_f32(x)
    Cn = x.shape[-1]
    rows = x.numel() // Cn
    y = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    check(lib.af_op_layernorm_fp8(ptr(x), ptr(_dev_f32(weight)), ptr(_dev_f32(bias)), eps, ptr(y), rows, Cn, act_shift,
                                  stream_ptr()), "af_op_layernorm_fp8")
    return y


def linear(x, weight, bias=None, residual=None, geglu=False, dtype="bf16"):
    """F.linear over the last dim; geglu=True applies GEGLU (attention.py:32-45) to the projection."""
    lib = _lib.load()
    x, weight = _dev_f32(x), _dev_f32(weight)
    K = x.shape[-1]
    M = x.numel() // K
    rows = weight.shape[0]
    N = rows // 2 if geglu else rows
    y = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
    b = None if bias is None else _dev_f32(bias)
    r = None if residual is None else _dev_f32(residual)
    check(lib.af_op_linear(DTYPES[dtype], ptr(x), ptr(weight), ptr(b), ptr(r), ptr(y), M, K, N, 1 if geglu else 0,
                           stream_ptr()), "af_op_linear")
    return y


def group_norm(x, weight, bias, eps=1e-5, silu=False, dtype="bf16"):
    """F.group_norm(x, 32, weight, bias, eps) (+ SiLU)."""
    lib = _lib.load()
    x = _dev_f32(x)
    B, Cn, H, W = x.shape
    y = torch.empty_like(x)
    check(lib.af_op_groupnorm(DTYPES[dtype], ptr(x), ptr(_dev_f32(weight)), ptr(_dev_f32(bias)), eps, 1 if silu else 0,
                              ptr(y), B, Cn, H, W, stream_ptr()), "af_op_groupnorm")
    return y


def conv_gn(x, w, b, gamma, beta, eps=1e-5, silu=True, residual=None):
    """conv3x3 (stride 1, bf16) then GroupNorm(32)(+SiLU) with the statistics summed in the convolution's epilogue
    (af_op_conv_gn).  Returns (conv output, GroupNorm output)."""
    lib = _lib.load()
    x = _dev_f32(x)
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    h = torch.empty(B, Cout, H, W, device=x.device, dtype=torch.float32)
    y = torch.empty_like(h)
    check(lib.af_op_conv_gn(ptr(x), ptr(_dev_f32(w)), ptr(_dev_f32(b)) if b is not None else None,
                            ptr(_dev_f32(residual)) if residual is not None else None, ptr(_dev_f32(gamma)), ptr(_dev_f32(beta)),
                            eps, 1 if silu else 0, ptr(h), ptr(y), B, Cin, H, W, Cout, stream_ptr()), "af_op_conv_gn")
    return h, y


def gn_conv1x1(x, gamma, beta, w, bias=None, eps=1e-6):
    """GroupNorm(32) + 1x1 convolution (SpatialTransformer.norm + proj_in) in bf16, both ways (af_op_gn_conv1x1): returns
    (plain, fused) = (apply pass + GEMM, row-panel GEMM with the GroupNorm in its prologue), each [B, N, H, W]."""
    lib = _lib.load()
    x = _dev_f32(x)
    B, Cn, H, W = x.shape
    N = w.shape[0]
    y0 = torch.empty(B, N, H, W, device=x.device, dtype=torch.float32)
    y1 = torch.empty_like(y0)
    check(lib.af_op_gn_conv1x1(ptr(x), ptr(_dev_f32(gamma)), ptr(_dev_f32(beta)), eps, ptr(_dev_f32(w.reshape(N, Cn))),
                               ptr(_dev_f32(bias)) if bias is not None else None, ptr(y0), ptr(y1), B, Cn, H, W, N, stream_ptr()),
          "af_op_gn_conv1x1")
    return y0, y1


def layer_norm(x, weight, bias, eps=1e-5, dtype="bf16"):
    lib = _lib.load()
    x = _dev_f32(x)
    Cn = x.shape[-1]
    y = torch.empty_like(x)
    check(lib.af_op_layernorm(DTYPES[dtype], ptr(x), ptr(_dev_f32(weight)), ptr(_dev_f32(bias)), eps, ptr(y),
                              x.numel() // Cn, Cn, stream_ptr()), "af_op_layernorm")
    return y


def attention(q, k, v, heads, scale=None, dtype="bf16", causal=False, nan_guard=False):
    """softmax(q k^T * scale) v per head; q [B,N,heads*dh], k/v [B,S,heads*dh] (attention.py:197-243).  causal: query i
    attends to keys <= i (the CLIP text tower's mask).  nan_guard (tests): K / V sit in front of 128 rows of NaNs."""
    lib = _lib.load()
    q, k, v = _dev_f32(q), _dev_f32(k), _dev_f32(v)
    B, Nq, Cn = q.shape
    Nk = k.shape[1]
    dh = Cn // heads
    scale = dh ** -0.5 if scale is None else scale
    o = torch.empty_like(q)
    check(lib.af_op_attention(DTYPES[dtype], ptr(q), ptr(k), ptr(v), ptr(o), B, Nq, Nk, heads, dh, scale,
                              (1 if causal else 0) | (2 if nan_guard else 0), stream_ptr()),
          "af_op_attention")
    return o


def xattn_fused(x, gamma, beta, wq, kv, wo, bo, eps=1e-5):
    """One cross-attention layer of a 64x64-level BasicTransformerBlock in ONE kernel (bf16; attention.py:172-257, 279):
    y = x + to_out(softmax(to_q(LayerNorm(x)) K^T / sqrt(40)) V) with x [B, N, 320], kv [B, S, 640] = the context's K | V
    projections.  Returns (y, parts) with parts [4, B * N, 2] = the LayerNorm partial sums (sum, sum of squares) of the stored
    rows, as the next consumer reads them."""
    lib = _lib.load()
    x = _dev_f32(x)
    B, N, Cn = x.shape
    S = kv.shape[1]
    xb = x.to(torch.bfloat16).float().reshape(B * N, Cn)            # statistics of the rows the kernel reads
    mu = xb.mean(dim=1)
    var = (xb * xb).mean(dim=1) - mu * mu
    stats = torch.stack([mu, torch.rsqrt(var.clamp_min(0) + eps)], dim=1).contiguous()
    y = torch.empty_like(x)
    parts = torch.empty(4, B * N, 2, device=x.device, dtype=torch.float32)
    check(lib.af_op_xattn_fused(ptr(x), ptr(stats), ptr(_dev_f32(gamma)), ptr(_dev_f32(beta)), ptr(_dev_f32(wq)), ptr(_dev_f32(kv)),
                                ptr(_dev_f32(wo)), ptr(_dev_f32(bo)), ptr(y), ptr(parts), B, N, S, stream_ptr()), "af_op_xattn_fused")
    return y, parts


def timestep_embedding(t, dim, dtype="f32"):
    lib = _lib.load()
    t = t.contiguous().long()
    y = torch.empty(t.shape[0], dim, device=t.device, dtype=torch.float32)
    check(lib.af_op_timestep_embedding(DTYPES[dtype], ptr(t), ptr(y), t.shape[0], dim, stream_ptr()),
          "af_op_timestep_embedding")
    return y


def ddim_step(x, e_cond, e_uncond, guidance, a_t, a_prev, sqrt_one_minus_at, sigma_t=0.0, noise=None, temperature=1.0):
    """CFG combine + DDIM update (ddim.py:260,279-295); returns (x_prev, pred_x0)."""
    lib = _lib.load()
    x, e_cond = _dev_f32(x), _dev_f32(e_cond)
    eu = None if e_uncond is None else _dev_f32(e_uncond)
    nz = None if noise is None else _dev_f32(noise)
    x_prev, pred_x0 = torch.empty_like(x), torch.empty_like(x)
    check(lib.af_ddim_step(ptr(x), ptr(e_cond), ptr(eu), ptr(nz), x.numel(), float(guidance), float(a_t), float(a_prev),
                           float(sqrt_one_minus_at), float(sigma_t), float(temperature), ptr(x_prev), ptr(pred_x0),
                           stream_ptr()), "af_ddim_step")
    return x_prev, pred_x0


def lincomb(terms, cfg=False):
    """sum_i w_i * x_i over up to four (tensor, weight) pairs; cfg=True: terms = [(e_cond, g), (e_uncond, _)] ->
    e_uncond + g * (e_cond - e_uncond)."""
    lib = _lib.load()
    xs = [_dev_f32(t) for t, _ in terms]
    ws = [float(w) for _, w in terms]
    while len(xs) < 4:
        xs.append(None)
        ws.append(0.0)
    out = torch.empty_like(xs[0])
    check(lib.af_lincomb(ptr(out), out.numel(), ptr(xs[0]), ws[0], ptr(xs[1]), ws[1], ptr(xs[2]), ws[2], ptr(xs[3]), ws[3],
                         1 if cfg else 0, stream_ptr()), "af_lincomb")
    return out


def to_uint8(img):
    """clamp((img+1)/2,0,1)*255 -> uint8 HWC (stable_txt2img.py:715,764-765)."""
    lib = _lib.load()
    img = _dev_f32(img)
    B, Cn, H, W = img.shape
    if Cn != 3:
        raise ValueError("to_uint8 expects [B,3,H,W]")
    y = torch.empty(B, H, W, 3, device=img.device, dtype=torch.uint8)
    check(lib.af_to_uint8(ptr(img), ptr(y), B, H, W, stream_ptr()), "af_to_uint8")
    return y
