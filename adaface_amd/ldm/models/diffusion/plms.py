"""Drop-in for ldm.models.diffusion.plms.PLMSSampler (reference plms.py:11-253), the `--plms` alternative of
scripts/stable_txt2img.py.  Same constructor / sample / plms_sampling / p_sample_plms signatures.  Note the
reference's quirks, kept: classifier-free guidance concatenates (UNCOND, COND) — the opposite order of the DDIM
sampler (plms.py:193-199) — takes `unconditional_guidance_scale` (a scalar, no annealing), and eta must be 0.
All tensor arithmetic runs in HIP kernels (af_lincomb, af_ddim_step); the sampler itself is host logic.
"""
from __future__ import annotations

import numpy as np
import torch

from adaface_amd import ops
from adaface_amd.ldm.modules.diffusionmodules.util import (make_ddim_sampling_parameters, make_ddim_timesteps,
                                                            noise_like)


class PLMSSampler(object):
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
        if ddim_eta != 0:
            raise ValueError('ddim_eta must be 0 for PLMS')
        self.ddim_timesteps = make_ddim_timesteps(ddim_discr_method=ddim_discretize, num_ddim_timesteps=ddim_num_steps,
                                                  num_ddpm_timesteps=self.ddpm_num_timesteps, verbose=verbose)
        acp = self.model.alphas_cumprod.detach().float().cpu()
        assert acp.shape[0] == self.ddpm_num_timesteps, 'alphas have to be defined for each timestep'
        to_dev = lambda x: x.clone().detach().to(torch.float32).to(self.model.device)
        self.register_buffer('betas', to_dev(self.model.betas))
        self.register_buffer('alphas_cumprod', to_dev(self.model.alphas_cumprod))
        self.register_buffer('alphas_cumprod_prev', to_dev(self.model.alphas_cumprod_prev))
        self.register_buffer('sqrt_one_minus_alphas_cumprod', to_dev(np.sqrt(1. - acp)))
        sig, a, a_prev = make_ddim_sampling_parameters(alphacums=acp, ddim_timesteps=self.ddim_timesteps, eta=ddim_eta,
                                                       verbose=verbose)
        self.ddim_sigmas, self.ddim_alphas, self.ddim_alphas_prev = sig, a, a_prev
        self.ddim_sqrt_one_minus_alphas = np.sqrt(1. - a)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, **kwargs):
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        return self.plms_sampling(conditioning, (batch_size, C, H, W), callback=callback, img_callback=img_callback,
                                  quantize_denoised=quantize_x0, mask=mask, x0=x0, ddim_use_original_steps=False,
                                  noise_dropout=noise_dropout, temperature=temperature, score_corrector=score_corrector,
                                  corrector_kwargs=corrector_kwargs, x_T=x_T, log_every_t=log_every_t,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning)

    @torch.no_grad()
    def plms_sampling(self, cond, shape, x_T=None, ddim_use_original_steps=False, callback=None, timesteps=None,
                      quantize_denoised=False, mask=None, x0=None, img_callback=None, log_every_t=100, temperature=1.,
                      noise_dropout=0., score_corrector=None, corrector_kwargs=None, unconditional_guidance_scale=1.,
                      unconditional_conditioning=None):
        """plms.py:119-173."""
        if ddim_use_original_steps:
            raise NotImplementedError("ddim_use_original_steps is not used by txt2img")
        device = self.model.betas.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T
        if timesteps is None:
            timesteps = self.ddim_timesteps
        else:
            subset_end = int(min(timesteps / self.ddim_timesteps.shape[0], 1) * self.ddim_timesteps.shape[0]) - 1
            timesteps = self.ddim_timesteps[:subset_end]
        intermediates = {'x_inter': [img], 'pred_x0': [img]}
        time_range = np.flip(timesteps)
        total_steps = timesteps.shape[0]
        old_eps = []
        self._twin_cache = None
        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            ts_next = torch.full((b,), int(time_range[min(i + 1, len(time_range) - 1)]), device=device, dtype=torch.long)
            if mask is not None:
                assert x0 is not None
                img = self.model.q_sample(x0, ts) * mask + (1. - mask) * img
            img, pred_x0, e_t = self.p_sample_plms(img, cond, ts, index=index, quantize_denoised=quantize_denoised,
                                                   temperature=temperature, noise_dropout=noise_dropout,
                                                   score_corrector=score_corrector, corrector_kwargs=corrector_kwargs,
                                                   unconditional_guidance_scale=unconditional_guidance_scale,
                                                   unconditional_conditioning=unconditional_conditioning,
                                                   old_eps=old_eps, t_next=ts_next)
            old_eps.append(e_t)
            if len(old_eps) >= 4:
                old_eps.pop(0)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates['x_inter'].append(img)
                intermediates['pred_x0'].append(pred_x0)
        self._twin_cache = None
        return img, intermediates

    def _twin_condition(self, c, uc):
        """(UNCOND, COND) — the reference's PLMS order (plms.py:193-199) — concatenated once per sampling run."""
        key = (id(c), id(uc))
        if self._twin_cache is not None and self._twin_cache[0] == key:
            return self._twin_cache[1]
        if isinstance(c, tuple):
            c_c, c_in_c, info = c
            c_u, c_in_u, _ = uc
            twin = (torch.cat([c_u, c_c]), sum([list(c_in_u), list(c_in_c)], []), info)
        else:
            twin = torch.cat([uc, c])
        self._twin_cache = (key, twin, c, uc)
        return twin

    @torch.no_grad()
    def p_sample_plms(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, old_eps=None, t_next=None):
        """plms.py:175-253."""
        if quantize_denoised or score_corrector is not None or use_original_steps:
            raise NotImplementedError("quantize_denoised / score_corrector / use_original_steps are not on the txt2img path")
        b = x.shape[0]
        f32 = lambda v: float(np.float32(float(v)))

        def get_model_output(xx, tt):
            if unconditional_conditioning is None or unconditional_guidance_scale == 1.:
                return self.model.apply_model(xx, tt, c)
            twin = self._twin_condition(c, unconditional_conditioning)
            if hasattr(self.model, "apply_model_cfg_twin"):
                e = self.model.apply_model_cfg_twin(xx, tt, twin)   # [x; x] without the concatenation (af_unet_forward_twin)
            else:
                e = self.model.apply_model(torch.cat([xx] * 2), torch.cat([tt] * 2), twin)
            return ops.lincomb([(e[b:], unconditional_guidance_scale), (e[:b], 0.0)], cfg=True)  # e_u + g (e_c - e_u)

        def get_x_prev_and_pred_x0(e, idx):
            # one noise_like draw per call, as the reference (plms.py:236), so the device generator advances
            # identically even when sigma_t = 0 (eta = 0) makes the term vanish
            sigma_t = f32(self.ddim_sigmas[idx])
            noise = noise_like(x.shape, x.device, repeat_noise)
            if noise_dropout > 0.:
                noise = torch.nn.functional.dropout(noise, p=noise_dropout)
            return ops.ddim_step(x, e, None, 1.0, f32(self.ddim_alphas[idx]), f32(self.ddim_alphas_prev[idx]),
                                 f32(self.ddim_sqrt_one_minus_alphas[idx]), sigma_t,
                                 None if sigma_t == 0. else noise, temperature)

        e_t = get_model_output(x, t)
        if len(old_eps) == 0:        # pseudo improved Euler (2nd order)
            x_prev, _ = get_x_prev_and_pred_x0(e_t, index)
            e_t_next = get_model_output(x_prev, t_next)
            e_t_prime = ops.lincomb([(e_t, 0.5), (e_t_next, 0.5)])
        elif len(old_eps) == 1:      # 2nd-order Adams-Bashforth
            e_t_prime = ops.lincomb([(e_t, 3 / 2), (old_eps[-1], -1 / 2)])
        elif len(old_eps) == 2:      # 3rd order
            e_t_prime = ops.lincomb([(e_t, 23 / 12), (old_eps[-1], -16 / 12), (old_eps[-2], 5 / 12)])
        else:                        # 4th order
            e_t_prime = ops.lincomb([(e_t, 55 / 24), (old_eps[-1], -59 / 24), (old_eps[-2], 37 / 24), (old_eps[-3], -9 / 24)])
        x_prev, pred_x0 = get_x_prev_and_pred_x0(e_t_prime, index)
        return x_prev, pred_x0, e_t
