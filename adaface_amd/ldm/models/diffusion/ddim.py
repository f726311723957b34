"""Drop-in for ldm.models.diffusion.ddim.DDIMSampler (reference ddim.py:12-350).

Same constructor, method names and keyword arguments.  Differences are confined to what
makes it MI355X-native and device-agnostic:
  * register_buffer does not force "cuda" (the reference does, ddim.py:22-26);
  * the CFG combine and the x_{t-1} update are ONE fused HIP kernel (af_ddim_step);
  * the (cond, uncond) context pair is concatenated once per sample() call, not once per
    step, so the UNet's hoisted cross-attention K/V stay cached across the 50 steps;
  * a scalar guidance_scale means "no annealing" instead of the reference's
    UnboundLocalError (ddim.py:169-173).
"""
from __future__ import annotations

import numpy as np
import torch

from adaface_amd import ops
from adaface_amd.ldm.modules.diffusionmodules.util import (extract_into_tensor, make_ddim_sampling_parameters,
                                                           make_ddim_timesteps, noise_like)


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        super().__init__()
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self._twin_cache = None

    def register_buffer(self, name, attr):
        if isinstance(attr, torch.Tensor) and attr.device != self.model.device:
            attr = attr.to(self.model.device)
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        """ddim.py:28-68."""
        self.ddim_timesteps = make_ddim_timesteps(ddim_discr_method=ddim_discretize, num_ddim_timesteps=ddim_num_steps,
                                                  num_ddpm_timesteps=self.ddpm_num_timesteps, verbose=verbose)
        alphas_cumprod = self.model.alphas_cumprod
        assert alphas_cumprod.shape[0] == self.ddpm_num_timesteps, 'alphas have to be defined for each timestep'
        to_torch = lambda x: x.clone().detach().to(torch.float32).to(self.model.device)
        acp_cpu = alphas_cumprod.detach().float().cpu()

        self.register_buffer('betas', to_torch(self.model.betas))
        self.register_buffer('alphas_cumprod', to_torch(alphas_cumprod))
        self.register_buffer('alphas_cumprod_prev', to_torch(self.model.alphas_cumprod_prev))
        self.register_buffer('sqrt_alphas_cumprod', to_torch(np.sqrt(acp_cpu)))
        self.register_buffer('sqrt_one_minus_alphas_cumprod', to_torch(np.sqrt(1. - acp_cpu)))
        self.register_buffer('log_one_minus_alphas_cumprod', to_torch(np.log(1. - acp_cpu)))
        self.register_buffer('sqrt_recip_alphas_cumprod', to_torch(np.sqrt(1. / acp_cpu)))
        self.register_buffer('sqrt_recipm1_alphas_cumprod', to_torch(np.sqrt(1. / acp_cpu - 1)))

        ddim_sigmas, ddim_alphas, ddim_alphas_prev = make_ddim_sampling_parameters(
            alphacums=acp_cpu, ddim_timesteps=self.ddim_timesteps, eta=ddim_eta, verbose=verbose)
        # the per-step scalars stay on the host: each is read once per step as a kernel argument
        self.ddim_sigmas = ddim_sigmas
        self.ddim_alphas = ddim_alphas
        self.ddim_alphas_prev = ddim_alphas_prev
        self.ddim_sqrt_one_minus_alphas = np.sqrt(1. - ddim_alphas)
        acp_prev = self.model.alphas_cumprod_prev.detach().float().cpu()
        self.ddim_sigmas_for_original_num_steps = ddim_eta * torch.sqrt(
            (1 - acp_prev) / (1 - acp_cpu) * (1 - acp_cpu / acp_prev))

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, guidance_scale=1.,
               unconditional_conditioning=None, **kwargs):
        """ddim.py:71-132."""
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        size = (batch_size, C, H, W)
        if verbose:
            print(f'Data shape for DDIM sampling is {size}, eta {eta}')
        return self.ddim_sampling(conditioning, size, callback=callback, img_callback=img_callback,
                                  quantize_denoised=quantize_x0, mask=mask, x0=x0, ddim_use_original_steps=False,
                                  noise_dropout=noise_dropout, temperature=temperature,
                                  score_corrector=score_corrector, corrector_kwargs=corrector_kwargs, x_T=x_T,
                                  log_every_t=log_every_t, guidance_scale=guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, verbose=verbose, **kwargs)

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, ddim_use_original_steps=False, callback=None, timesteps=None,
                      quantize_denoised=False, mask=None, x0=None, img_callback=None, log_every_t=100,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      guidance_scale=1., unconditional_conditioning=None, verbose=False, **kwargs):
        """ddim.py:135-220: the S-iteration loop with annealed guidance (:169-180,215-218)."""
        device = self.model.betas.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T
        if timesteps is None:
            timesteps = self.ddpm_num_timesteps if ddim_use_original_steps else self.ddim_timesteps
        elif not ddim_use_original_steps:
            subset_end = int(min(timesteps / self.ddim_timesteps.shape[0], 1) * self.ddim_timesteps.shape[0]) - 1
            timesteps = self.ddim_timesteps[:subset_end]
        intermediates = {'x_inter': [img], 'pred_x0': [img]}
        time_range = reversed(range(0, timesteps)) if ddim_use_original_steps else np.flip(timesteps)
        total_steps = timesteps if ddim_use_original_steps else timesteps.shape[0]
        if verbose:
            print(f"Running DDIM Sampling with {total_steps} timesteps")

        if isinstance(guidance_scale, (list, tuple)):
            max_guide_scale, min_guide_scale = guidance_scale
        else:
            max_guide_scale = min_guide_scale = guidance_scale
        max_guide_anneal_steps = total_steps - 1
        delta = (max_guide_scale - min_guide_scale) / max_guide_anneal_steps if max_guide_anneal_steps > 0 else 0.
        guide_scale = max_guide_scale
        self._twin_cache = None

        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            if mask is not None:
                assert x0 is not None
                img_orig = self.model.q_sample(x0, ts)
                img = img_orig * mask + (1. - mask) * img
            img, pred_x0 = self.p_sample_ddim(img, cond, ts, index=index, use_original_steps=ddim_use_original_steps,
                                              quantize_denoised=quantize_denoised, temperature=temperature,
                                              noise_dropout=noise_dropout, score_corrector=score_corrector,
                                              corrector_kwargs=corrector_kwargs, guidance_scale=guide_scale,
                                              unconditional_conditioning=unconditional_conditioning)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates['x_inter'].append(img)
                intermediates['pred_x0'].append(pred_x0)
            guide_scale = guide_scale - delta if i <= max_guide_anneal_steps else 1
        self._twin_cache = None
        return img, intermediates

    def _twin_condition(self, c, uc):
        """(cond, uncond) concatenated in the reference's order — cond FIRST (ddim.py:236-247) — built once."""
        key = (id(c), id(uc))
        if self._twin_cache is not None and self._twin_cache[0] == key:
            return self._twin_cache[1]
        if isinstance(c, tuple):
            c_c, c_in_c, extra_info = c
            c_u, c_in_u, _ = uc
            twin = (torch.cat([c_c, c_u]), sum([list(c_in_c), list(c_in_u)], []), extra_info)
        else:
            twin = torch.cat([c, uc])
        self._twin_cache = (key, twin, c, uc)  # keep c / uc alive so the ids stay unique
        return twin

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      guidance_scale=1., unconditional_conditioning=None):
        """ddim.py:222-296."""
        b, device = x.shape[0], x.device
        if unconditional_conditioning is None or guidance_scale == 1.:
            e_c, e_u = self.model.apply_model(x, t, c), None
        else:
            twin = self._twin_condition(c, unconditional_conditioning)
            if hasattr(self.model, "apply_model_cfg_twin"):
                # [x; x] without the concatenation: the UNet computes its context-independent prefix once (af_unet_forward_twin)
                e = self.model.apply_model_cfg_twin(x, t, twin)
            else:
                e = self.model.apply_model(torch.cat([x] * 2), torch.cat([t] * 2), twin)
            e_c, e_u = e[:b], e[b:]

        alphas = self.model.alphas_cumprod if use_original_steps else self.ddim_alphas
        alphas_prev = self.model.alphas_cumprod_prev if use_original_steps else self.ddim_alphas_prev
        sqrt_one_minus_alphas = self.model.sqrt_one_minus_alphas_cumprod if use_original_steps else self.ddim_sqrt_one_minus_alphas
        sigmas = self.ddim_sigmas_for_original_num_steps if use_original_steps else self.ddim_sigmas
        # the reference materialises each scalar through torch.full(...) => fp32 (ddim.py:273-276)
        f32 = lambda v: float(np.float32(float(v)))
        sigma_t = f32(sigmas[index])
        # drawn every step like the reference (ddim.py:286) so the generator state advances identically,
        # even though sigma_t = 0 (eta = 0) makes the term vanish
        noise = noise_like(x.shape, device, repeat_noise)
        if noise_dropout > 0.:
            noise = torch.nn.functional.dropout(noise, p=noise_dropout)
        if sigma_t == 0.:
            noise = None
        if score_corrector is not None:
            # ddim.py:262-264: the corrector sees the COMBINED score; combine first (af_lincomb, the kernel's own CFG form),
            # then hand its result to the update as a plain eps
            assert self.model.parameterization == "eps"
            e_t = e_c if e_u is None else ops.lincomb([(e_c, guidance_scale), (e_u, 0.0)], cfg=True)
            e_c, e_u = score_corrector.modify_score(self.model, e_t, x, t, c, **(corrector_kwargs or {})), None
        x_prev, pred_x0 = ops.ddim_step(x, e_c, e_u, guidance_scale, f32(alphas[index]), f32(alphas_prev[index]),
                                        f32(sqrt_one_minus_alphas[index]), sigma_t, noise, temperature)
        if quantize_denoised:
            # ddim.py:281-282,293: x_prev is rebuilt around the quantised pred_x0 (a VQ first stage; AutoencoderKL has no
            # `quantize`, and the reference raises AttributeError there as this does).  x_prev - sqrt(a_prev) pred_x0 is the
            # dir_xt + noise term the fused kernel already formed, so only the pred_x0 term is swapped (fp32 scalars as the
            # reference's torch.full(...) materialises them)
            q, _, *_ = self.model.first_stage_model.quantize(pred_x0)
            sa = float(np.sqrt(np.float32(f32(alphas_prev[index]))))
            x_prev = ops.lincomb([(x_prev, 1.0), (pred_x0, -sa), (q, sa)])
            pred_x0 = q
        return x_prev, pred_x0

    @torch.no_grad()
    def stochastic_encode(self, x0, t, use_original_steps=False, noise=None):
        """ddim.py:299-312."""
        if use_original_steps:
            sqrt_alphas_cumprod = self.sqrt_alphas_cumprod
            sqrt_one_minus_alphas_cumprod = self.sqrt_one_minus_alphas_cumprod
        else:
            sqrt_alphas_cumprod = torch.sqrt(self.ddim_alphas).to(x0.device)
            sqrt_one_minus_alphas_cumprod = torch.as_tensor(self.ddim_sqrt_one_minus_alphas).to(x0.device)
        if noise is None:
            noise = torch.randn_like(x0)
        return (extract_into_tensor(sqrt_alphas_cumprod, t, x0.shape) * x0 +
                extract_into_tensor(sqrt_one_minus_alphas_cumprod, t, x0.shape) * noise)

    @torch.no_grad()
    def decode(self, x_latent, cond, t_start, guidance_scale=1.0, unconditional_conditioning=None,
               use_original_steps=False):
        """ddim.py:315-350 (img2img tail): anneals guidance from `guidance_scale` to min(2, guidance_scale)."""
        timesteps = np.arange(self.ddpm_num_timesteps) if use_original_steps else self.ddim_timesteps
        timesteps = timesteps[:t_start]
        time_range = np.flip(timesteps)
        total_steps = timesteps.shape[0]
        max_g = guidance_scale
        min_g = min(2.0, max_g)
        delta = (max_g - min_g) / (total_steps - 1) if total_steps > 1 else 0.
        g = max_g
        x_dec = x_latent
        self._twin_cache = None
        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((x_latent.shape[0],), int(step), device=x_latent.device, dtype=torch.long)
            x_dec, _ = self.p_sample_ddim(x_dec, cond, ts, index=index, use_original_steps=use_original_steps,
                                          guidance_scale=g, unconditional_conditioning=unconditional_conditioning)
            g = g - delta
        self._twin_cache = None
        return x_dec
