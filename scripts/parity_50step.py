#!/usr/bin/env python
"""North-star parity check at full size: SD-1.5, one sample with CFG (Bf = 2), 50 DDIM steps, annealed guidance [10, 4],
same seeded weights / x_T / contexts through (a) the drop-in DDIMSampler on the GPU in f32 (parity) mode and in bf16 mode
and (b) the CPU oracle (fp32 torch restatement of the reference).  Prints the max-abs and relative difference of the FINAL
latent.  ~4 minutes of CPU time for the oracle's 50 forwards; not part of the pytest suite for that reason.

    python scripts/parity_50step.py [--steps 50]
"""
import argparse
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import ldm_oracle as O  # noqa: E402  (checker)

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--threads", type=int, default=16)
args = ap.parse_args()
torch.set_num_threads(args.threads)
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")

from adaface_amd.configs import sd15_config  # noqa: E402
from ldm.models.diffusion.ddim import DDIMSampler  # noqa: E402
from ldm.util import instantiate_from_config  # noqa: E402

sd = O.synth_state_dict(O.unet_param_shapes(O.SD15_UNET), seed=21)
model = instantiate_from_config(sd15_config()["model"]).eval()
missing, unexpected = model.load_state_dict(sd, strict=False)
assert not unexpected
model = model.to(dev)
g = torch.Generator().manual_seed(99)
x_T = torch.randn(1, 4, 64, 64, generator=g)
c = torch.randn(16, 77, 768, generator=g)
uc = torch.randn(16, 77, 768, generator=g)
S = args.steps


def gpu_run(mode):
    model.set_compute_dtype(mode)
    sampler = DDIMSampler(model)
    t0 = time.perf_counter()
    z, _ = sampler.sample(S=S, conditioning=model.get_learned_conditioning(c.to(dev)), batch_size=1, shape=[4, 64, 64],
                          verbose=False, guidance_scale=[10.0, 4.0],
                          unconditional_conditioning=model.get_learned_conditioning(uc.to(dev)), eta=0.0, x_T=x_T.to(dev))
    torch.cuda.synchronize()
    return z.cpu(), time.perf_counter() - t0


z32, t32 = gpu_run("f32")
z16, t16 = gpu_run("bf16")
print(f"GPU f32 mode: {t32:.1f} s, bf16 mode: {t16:.1f} s for {S} steps", flush=True)
t0 = time.perf_counter()
n = [0]


def apply(x, t, ctx):
    n[0] += 1
    if n[0] % 10 == 0:
        print(f"  oracle forward {n[0]} / {S}  ({time.perf_counter() - t0:.0f} s)", flush=True)
    return O.unet_forward(sd, O.SD15_UNET, x, t, ctx)


ref = O.ddim_sample(apply, O.register_schedule(), S, x_T, c, uc, guidance_scale=(10.0, 4.0))
print(f"CPU oracle: {time.perf_counter() - t0:.0f} s on {args.threads} threads")
scale = ref.abs().max().item()
for name, z in (("f32", z32), ("bf16", z16)):
    d = (z - ref).abs().max().item()
    print(f"final latent after {S} DDIM steps, {name} mode vs CPU oracle: max-abs diff {d:.3e}  (max|ref| {scale:.3f}, relative {d / scale:.3e})")
