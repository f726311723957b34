#!/usr/bin/env python
"""Where does a bench step's wall time go?  50 bare UNet forwards vs the sampler loop vs the VAE decode (bf16, Bf=16)."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from adaface_amd.synth import synth_context  # noqa: E402
from ldm.models.diffusion.ddim import DDIMSampler  # noqa: E402

dev = torch.device("cuda:0")
model = bench.build_model(dev, "bf16")
sampler = DDIMSampler(model)
B = 8
x_T = torch.randn(B, 4, 64, 64, device=dev)
c = model.get_learned_conditioning(synth_context(B, seed=100, device=dev))
uc = model.get_learned_conditioning(synth_context(B, seed=101, device=dev, shared=True))


def sample():
    return sampler.sample(S=50, conditioning=c, batch_size=B, shape=[4, 64, 64], verbose=False, guidance_scale=[10.0, 4.0],
                          unconditional_conditioning=uc, eta=0.0, x_T=x_T)[0]


def timed(fn, n=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


t_s, z = timed(sample)
t_d, _ = timed(lambda: model.decode_first_stage_uint8(z))
eng = model.model.diffusion_model.engine(dev)
x = torch.randn(16, 4, 64, 64, device=dev)
t = torch.full((16,), 500, device=dev, dtype=torch.long)
out = torch.empty_like(x)
t_f, _ = timed(lambda: [eng.unet_forward(x, t, out) for _ in range(50)])
print(f"sampler.sample 50 steps: {t_s:.1f} ms   50 bare forwards: {t_f:.1f} ms   decode+uint8: {t_d:.1f} ms")
