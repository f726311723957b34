"""Host-side schedule helpers with the reference's names and argument meaning
(ldm/modules/diffusionmodules/util.py).  Pure host logic: tables of 50-1000 scalars
computed once per sample() call; nothing here runs per denoising step.
"""
from __future__ import annotations

import numpy as np
import torch


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """util.py:21-43.  Only the schedules a v1 inference config can select are provided."""
    if schedule == "linear":
        betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2
    elif schedule == "sqrt_linear":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64)
    elif schedule == "sqrt":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64) ** 0.5
    elif schedule == "cosine":
        steps = torch.arange(n_timestep + 1, dtype=torch.float64) / n_timestep + cosine_s
        alphas = torch.cos(steps / (1 + cosine_s) * np.pi / 2).pow(2)
        alphas = alphas / alphas[0]
        betas = (1 - alphas[1:] / alphas[:-1]).clamp(min=0, max=0.999)
    else:
        raise ValueError(f"schedule '{schedule}' unknown.")
    return betas.numpy()


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    """util.py:46-60: 'uniform' -> arange(0, T, T // S) + 1; 'quad' -> (linspace(0, sqrt(.8T), S))^2 + 1."""
    if ddim_discr_method == "uniform":
        c = num_ddpm_timesteps // num_ddim_timesteps
        ddim_timesteps = np.asarray(list(range(0, num_ddpm_timesteps, c)))
    elif ddim_discr_method == "quad":
        ddim_timesteps = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps)) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    steps_out = ddim_timesteps + 1
    if verbose:
        print(f"Selected timesteps for ddim sampler: {steps_out}")
    return steps_out


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """util.py:63-77.  alphacums: fp32 tensor (CPU).  Returns (sigmas, alphas, alphas_prev) with the
    reference's mixed types: alphas a torch tensor, alphas_prev / sigmas numpy-backed."""
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    if verbose:
        print(f"Selected alphas for ddim sampler: a_t: {alphas}; a_(t-1): {alphas_prev}")
        print(f"For the chosen value of eta, which is {eta}, this results in sigma_t schedule {sigmas}")
    return sigmas, alphas, alphas_prev


def noise_like(shape, device, repeat=False):
    """util.py:267-270."""
    if repeat:
        return torch.randn((1, *shape[1:]), device=device).repeat(shape[0], *((1,) * (len(shape) - 1)))
    return torch.randn(shape, device=device)


def extract_into_tensor(a, t, x_shape):
    """util.py:80-83: gather a[t] and reshape to broadcast over x."""
    b = t.shape[0]
    out = a.gather(-1, t)
    return out.reshape(b, *((1,) * (len(x_shape) - 1)))


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False, dtype="f32"):
    """util.py:154-174 through the HIP kernel (device tensors only)."""
    from adaface_amd import ops
    if max_period != 10000 or repeat_only:
        raise NotImplementedError("only max_period=10000, repeat_only=False is on the hot path")
    return ops.timestep_embedding(timesteps, dim, dtype=dtype)
