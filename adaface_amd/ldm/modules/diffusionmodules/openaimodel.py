"""Drop-in for ldm.modules.diffusionmodules.openaimodel.UNetModel (reference
openaimodel.py:417-1052): same constructor kwargs, same forward signature, same state_dict
keys; the forward pass is one af_unet_forward call into the hand-written HIP path.
"""
from __future__ import annotations

from typing import Optional

import torch

from adaface_amd import layout
from adaface_amd.ldm._hipmodule import HipModule, build_param_tree


class UNetModel(HipModule):
    _ckpt_prefix = "model.diffusion_model."

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None,
                 use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False,
                 use_spatial_transformer=False, transformer_depth=1, context_dim=None, n_embed=None, legacy=True):
        super().__init__()
        # the SD-v1 family is the hot path (v1-inference-ada.yaml:35-50); other branches of the
        # reference constructor are training / legacy-LDM variants outside SURVEY.md §8
        unsupported = {
            "dims != 2": dims != 2, "num_classes": num_classes is not None, "use_scale_shift_norm": use_scale_shift_norm,
            "resblock_updown": resblock_updown, "n_embed": n_embed is not None, "dropout": dropout != 0,
            "not conv_resample": not conv_resample, "not use_spatial_transformer": not use_spatial_transformer,
            "num_head_channels": num_head_channels != -1,
            "num_heads_upsample": num_heads_upsample not in (-1, num_heads),
        }
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError(f"UNetModel: options outside the SD-v1 denoising path: {bad}")
        if context_dim is None or num_heads == -1:
            raise ValueError("UNetModel: context_dim and num_heads are required (spatial transformer)")
        if hasattr(context_dim, "__len__"):
            context_dim = list(context_dim)
            if len(context_dim) != 1:
                raise NotImplementedError("a single context_dim is supported")
            context_dim = context_dim[0]
        self.image_size = image_size
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = tuple(attention_resolutions)
        self.channel_mult = tuple(channel_mult)
        self.num_heads = num_heads
        self.context_dim = int(context_dim)
        self.transformer_depth = transformer_depth
        self.use_checkpoint = use_checkpoint  # accepted, meaningless at inference (util.py:115)
        self.dtype = torch.float32            # dtype of the tensors crossing the boundary
        self.debug_attn = False
        if use_fp16:
            self.compute_dtype = "bf16"
        shapes = layout.unet_param_shapes(**self._layout_kwargs())
        build_param_tree(self, shapes, zero_init=layout.unet_zero_init_names(shapes))
        object.__setattr__(self, "_ctx_key", None)

    def _layout_kwargs(self):
        return dict(in_channels=self.in_channels, model_channels=self.model_channels, out_channels=self.out_channels,
                    num_res_blocks=self.num_res_blocks, attention_resolutions=self.attention_resolutions,
                    channel_mult=self.channel_mult, context_dim=self.context_dim,
                    transformer_depth=self.transformer_depth)

    def _engine_kwargs(self):
        kw = self._layout_kwargs()
        kw.update(num_heads=self.num_heads, n_context_layers=16)
        return {"unet": kw}

    def _mark_dirty(self):
        super()._mark_dirty()
        object.__setattr__(self, "_ctx_key", None)  # cached K/V depend on the weights

    def _compel_cfg(self, context, B, info):
        """Inference-time compel cfg (openaimodel.py:898-916 + prob_apply_compel_cfg, util.py:2063-2094): per conditioned
        layer, with probability apply_compel_cfg_prob, the context of the FIRST half of the batch becomes
        (ctx - empty) * 1.1**level + empty.  The Python `random` draws are made in the reference's order (one per layer,
        then the level and two inner draws when it applies), so a seeded run re-weights the same layers.  Returns the
        re-weighted context and whether it may be cached (prob >= 1 and a degenerate level range: the reference
        re-draws on every forward otherwise).  The arithmetic is af_lincomb: w*ctx + (1-w)*empty."""
        import random
        from adaface_amd import ops
        empty, rng, prob = info.get("empty_context", None), info.get("compel_cfg_weight_level_range", None), info["apply_compel_cfg_prob"]
        if empty is None or rng is None:
            return context, True
        is_range = isinstance(rng, (list, tuple))
        deterministic = prob >= 1 and (not is_range or rng[0] == rng[1])
        key = ("compel", id(context), getattr(context, "_version", 0), id(empty), prob, tuple(rng) if is_range else rng)
        if deterministic and getattr(self, "_compel_key", None) == key:
            return self._compel_ctx, True
        L = 16
        ctx = context.reshape(B, L, -1, context.shape[-1]).clone()
        half = B // 2
        e = empty.to(ctx.device, torch.float32).expand(half, *ctx.shape[2:]).contiguous()
        for l in range(L):
            if random.random() > prob:
                continue
            level = random.uniform(*rng) if is_range else rng
            random.random(); random.random()          # the recursive calls on (v, k) draw once each (util.py:2077)
            w = 1.1 ** level
            if half > 0:
                ctx[:half, l] = ops.lincomb([(ctx[:half, l].contiguous(), w), (e, 1.0 - w)])
        ctx = ctx.reshape(context.shape)
        if deterministic:
            object.__setattr__(self, "_compel_key", key)
            object.__setattr__(self, "_compel_ctx", ctx)
            object.__setattr__(self, "_compel_refs", (context, empty))
        return ctx, deterministic

    @staticmethod
    def _conv_attn_spec(ks, placeholder2indices):
        """(ks, batch indices, token positions) for Engine.set_conv_attn from the reference's
        extra_info['placeholder2indices'] = {subject string: (indices_B, indices_N)} (util.py:711-727): the M >= ks*ks
        positions of each subject sample are consecutive in indices_N; the first ks*ks are used.  Every subject string of
        the dict contributes its own rows (attention.py:208-216 loops over them); a sample that carries several strings
        appears once per string, in the dict's order.  Kernel sizes 2, 3, 4 as the reference (util.py:747-760)."""
        if ks is None or ks <= 1 or not placeholder2indices:
            return (0, (), ())
        if ks not in (2, 3, 4):
            raise NotImplementedError(f"conv attention kernel size {ks}: the reference pads for 2, 3 and 4 only (util.py:747-760)")
        batch, tokens = [], []
        for indices in placeholder2indices.values():
            if indices is None:
                continue
            idx_b = [int(v) for v in torch.as_tensor(indices[0]).tolist()]
            idx_n = [int(v) for v in torch.as_tensor(indices[1]).tolist()]
            if not idx_b:
                continue
            uniq = sorted(set(idx_b))
            M = len(idx_n) // len(uniq)
            if M < ks * ks:
                raise ValueError(f"{M} embeddings are not enough to cover a {ks}x{ks} kernel")
            for i, bi in enumerate(uniq):
                batch.append(bi)
                tokens.append(tuple(idx_n[i * M: i * M + ks * ks]))
        return (ks, tuple(batch), tuple(tokens))

    @torch.no_grad()
    def forward(self, x, timesteps=None, context=None, y=None, context_in=None, extra_info=None, cfg_twin=False, **kwargs):
        """x [B,C,H,W], timesteps [B], context [B*16,T,D] (layerwise) or [B,T,D]; returns eps [B,C,H,W] fp32.
        Mirrors openaimodel.py:827-1052 for inference: `extra_info` keys read at :849-859.
        cfg_twin (not in the reference; used by this package's samplers): x / timesteps are ONE half of the
        classifier-free-guidance batch, the context is that of [x; x] (cond first, ddim.py:236-247); returns eps for the
        2B samples exactly as forward(torch.cat([x] * 2), torch.cat([t] * 2), context) would."""
        if y is not None:
            raise NotImplementedError("class-conditional UNet (num_classes) is not on the path")
        if timesteps is None or context is None:
            raise ValueError("UNetModel.forward needs timesteps and context")
        info = extra_info if extra_info is not None else {}
        layerwise = bool(info.get("use_layerwise_context", False))
        ks = info.get("use_conv_attn_kernel_size", None)
        conv = self._conv_attn_spec(ks, info.get("placeholder2indices", None))
        if info.get("img_mask", None) is not None or info.get("capture_distill_attn", False):
            raise NotImplementedError("img_mask / capture_distill_attn are training-time options")
        if info.get("iter_type", "normal_recon") == "mix_hijk":
            raise NotImplementedError("iter_type 'mix_hijk' (separate k/v contexts) is a training-time option")
        eng = self.engine(x.device)
        B = x.shape[0] * (2 if cfg_twin else 1)
        cache_ok = True
        if layerwise and info.get("apply_compel_cfg_prob", 0) > 0:
            context, cache_ok = self._compel_cfg(context.to(x.device), B, info)
        # Hoisted cross-attention K/V are cached per context TENSOR OBJECT: a reference to it is kept so that its
        # id / storage cannot be recycled for another prompt while the cache is live (data_ptr alone is unsafe).
        key = (id(context), getattr(context, "_version", 0), tuple(context.shape), layerwise, B, conv)
        if key != self._ctx_key or not cache_ok:
            eng.set_conv_attn(*conv)
            eng.set_context(context.to(x.device), B, layerwise)
            object.__setattr__(self, "_ctx_key", key if cache_ok else None)
            object.__setattr__(self, "_ctx_ref", context)
        out = (eng.unet_forward_twin if cfg_twin else eng.unet_forward)(x, timesteps.to(x.device))
        if extra_info is not None:
            # the reference writes the (here empty) distillation capture into the caller's dict (:1031-1035)
            extra_info["ca_layers_activations"] = {k: {} for k in ("outfeat", "attn", "attnscore", "q")}
        return out.type(x.dtype) if x.dtype != torch.float32 else out
