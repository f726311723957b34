"""Drop-in for ldm.models.autoencoder.AutoencoderKL (reference autoencoder.py:285-423),
decode side: post_quant_conv + Decoder (model.py:502-608) as one af_vae_decode call; encode side (init-image path,
SURVEY.md §8f-3): Encoder (model.py:408-499) + quant_conv as one af_vae_encode call returning the posterior.
"""
from __future__ import annotations

import torch

from adaface_amd import layout
from adaface_amd.ldm._hipmodule import HipModule, build_param_tree


class DiagonalGaussianDistribution:
    """ldm/modules/distributions/distributions.py:24-62 (the members the inference path reads).  `sample()` draws its
    noise like the reference does — torch.randn on the HOST generator, then moved to the parameters' device
    (distributions.py:36) — so a seeded init-image latent matches the reference's; the arithmetic is af_posterior_sample."""

    def __init__(self, parameters, deterministic=False, engine=None):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self._engine = engine

    @property
    def std(self):
        return torch.zeros_like(self.mean) if self.deterministic else torch.exp(0.5 * self.logvar)

    @property
    def var(self):
        return torch.zeros_like(self.mean) if self.deterministic else torch.exp(self.logvar)

    def sample(self, noise=None, scale: float = 1.0):
        if noise is None and not self.deterministic:
            noise = torch.randn(self.mean.shape).to(device=self.parameters.device)
        return self._engine.posterior_sample(self.parameters, None if self.deterministic else noise, scale)

    def mode(self):
        return self.mean


class AutoencoderKL(HipModule):
    _ckpt_prefix = "first_stage_model."

    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=[], image_key="image",
                 colorize_nlabels=None, monitor=None):
        super().__init__()
        dd = dict(ddconfig)
        if dd.get("attn_resolutions"):
            raise NotImplementedError("decoder attention at up-levels (attn_resolutions) is not in the SD-v1 VAE")
        if not dd.get("double_z", True):
            raise NotImplementedError("AutoencoderKL requires double_z")
        self.ddconfig = dd
        self.embed_dim = embed_dim
        self.image_key = image_key
        self.z_channels = dd["z_channels"]
        dec = {"decoder." + k: v for k, v in layout.vae_decoder_param_shapes(
            ch=dd["ch"], out_ch=dd["out_ch"], ch_mult=tuple(dd["ch_mult"]), num_res_blocks=dd["num_res_blocks"],
            z_channels=dd["z_channels"]).items()}
        dec["post_quant_conv.weight"] = (dd["z_channels"], embed_dim, 1, 1)
        dec["post_quant_conv.bias"] = (dd["z_channels"],)
        dec.update({"encoder." + k: v for k, v in layout.vae_encoder_param_shapes(
            ch=dd["ch"], ch_mult=tuple(dd["ch_mult"]), num_res_blocks=dd["num_res_blocks"], z_channels=dd["z_channels"],
            in_channels=dd.get("in_channels", 3)).items()})
        dec["quant_conv.weight"] = (2 * embed_dim, 2 * dd["z_channels"], 1, 1)
        dec["quant_conv.bias"] = (2 * embed_dim,)
        build_param_tree(self, dec)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)

    def _engine_kwargs(self):
        dd = self.ddconfig
        return {"vae": dict(ch=dd["ch"], out_ch=dd["out_ch"], ch_mult=tuple(dd["ch_mult"]),
                            num_res_blocks=dd["num_res_blocks"], z_channels=dd["z_channels"],
                            embed_dim=self.embed_dim, encoder=True, in_channels=dd.get("in_channels", 3))}

    def init_from_ckpt(self, path, ignore_keys=()):
        sd = torch.load(path, map_location="cpu", weights_only=True)
        sd = sd.get("state_dict", sd)
        sd = {k: v for k, v in sd.items() if not any(k.startswith(ik) for ik in ignore_keys)}
        self.load_state_dict(sd, strict=False)

    @torch.no_grad()
    def decode(self, z, scale_factor: float = 1.0, return_uint8: bool = False):
        """autoencoder.py:330-333.  `scale_factor` lets decode_first_stage fold its 1/0.18215 in."""
        eng = self.engine(z.device)
        if return_uint8:
            return eng.vae_decode(z, scale_factor=scale_factor, want_uint8=True, want_float=False)
        return eng.vae_decode(z, scale_factor=scale_factor)

    @torch.no_grad()
    def encode(self, x, mask=None):
        """autoencoder.py:324-328: Encoder + quant_conv -> DiagonalGaussianDistribution."""
        if mask is not None:
            raise NotImplementedError("AutoencoderKL.encode: attention masks are a training-time option")
        eng = self.engine(x.device)
        return DiagonalGaussianDistribution(eng.vae_encode(x), engine=eng)

    def forward(self, input, sample_posterior=True, mask=None):
        raise NotImplementedError("autoencoder training forward is out of scope")
