#!/usr/bin/env python
"""Headline benchmark: 512x512 images/sec @ 50 DDIM steps, batch 8 per GPU (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: DDIMSampler.sample (50 steps x CFG-doubled UNet forward
at Bf=16 on 4x64x64 latents with a [256,77,768] layerwise context) + decode_first_stage to 8 uint8 512x512
frames, through the drop-in classes (ldm.models.diffusion.ddim.DDIMSampler, LatentDiffusion.apply_model,
AutoencoderKL.decode) and the C ABI underneath.  Inputs are resident in HBM before the timed region.
Multi-GPU: independent samples, batch 8 per rank (weak scaling), ONE RCCL all-gather of the decoded uint8
frames per step.  Synthetic data and seeded random-init weights of the SD-1.5 architecture.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

F_UNET = 803.273e9      # algorithmic FLOP per sample-forward
F_VAE = 2514.519e9      # algorithmic FLOP per decoded image
PEAK_BF16 = 2.5e15      # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_F32 = 157.3e12
K_NAMES = ["conv_gemm", "attention", "groupnorm", "layernorm", "other"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the HIP-event roofline leg")
    ap.add_argument("--event-stride", type=int, default=7,
                    help="HIP-event bracket every n-th GEMM / attention launch inside the timed region (1 = all)")
    return ap.parse_args()


def build_model(device, dtype):
    from adaface_amd.configs import sd15_config
    from ldm.util import instantiate_from_config
    model = instantiate_from_config(sd15_config()["model"]).eval()
    model.set_compute_dtype(dtype)
    model = model.to(device)
    # seeded synthetic weights, drawn in HBM (identical on every rank)
    g = torch.Generator(device=device).manual_seed(1234)
    with torch.no_grad():
        for name, p in sorted(model.named_parameters()):
            if p.dim() == 1:
                t = torch.randn(p.shape, generator=g, device=device)
                p.copy_(1.0 + 0.1 * t if name.endswith(".weight") else 0.05 * t)
            else:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g, device=device) * fan_in ** -0.5)
    model.model.diffusion_model._mark_dirty()
    model.first_stage_model._mark_dirty()
    return model


def cpu_baseline(ddim_steps):
    """Oracle (CPU restatement of the reference, oracle/ldm_oracle.py) on the host cores: one UNet forward at
    CFG batch 2 (= 1/ddim_steps of one image's denoising) + one VAE decode, extrapolated to images/sec."""
    from oracle import ldm_oracle as O
    # one GPU's share of the host is 16 cores; more threads than that oversubscribes the box
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(7)
    sd = O.synth_state_dict(O.unet_param_shapes(O.SD15_UNET), seed=21)
    x = torch.randn(2, 4, 64, 64, generator=g)
    t = torch.tensor([981, 981])
    ctx = torch.randn(2 * 16, 77, 768, generator=g)
    with torch.no_grad():
        O.unet_forward(sd, O.SD15_UNET, x, t, ctx)            # warm-up (page-in, thread pool)
        t0 = time.perf_counter()
        for _ in range(2):
            O.unet_forward(sd, O.SD15_UNET, x, t, ctx)
        t_unet = (time.perf_counter() - t0) / 2
        del sd
        vsd = O.synth_state_dict(O.vae_param_shapes(O.SD15_VAE), seed=22)
        z = torch.randn(1, 4, 64, 64, generator=g) * 0.18215
        t0 = time.perf_counter()
        O.vae_decode(vsd, O.SD15_VAE, z)
        t_vae = time.perf_counter() - t0
    return {"value": 1.0 / (ddim_steps * t_unet + t_vae), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"UNet forward at CFG batch 2 (warm-up + mean of 2: {t_unet:.2f} s) + 1 VAE decode ({t_vae:.2f} s), fp32 torch CPU, "
                      f"extrapolated as 1/({ddim_steps}*t_unet + t_vae)"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: adaface_amd has no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=device)
    from adaface_amd import _lib
    from adaface_amd.parallel import gather_frames, shard_batch
    from adaface_amd.synth import synth_context
    from ldm.models.diffusion.ddim import DDIMSampler
    lib = _lib.load()

    B, S = args.batch, args.ddim_steps
    model = build_model(device, args.dtype)
    sampler = DDIMSampler(model)
    # global inputs generated from one seed on the host, sliced per rank (results independent of world size)
    g = torch.Generator().manual_seed(42)
    x_T = shard_batch(torch.randn(world * B, 4, 64, 64, generator=g), rank, world).to(device)
    c_emb = shard_batch(synth_context(world * B, seed=100, device="cpu"), rank, world, per_sample=16).to(device)
    uc_emb = synth_context(B, seed=101, device=device, shared=True)
    c = model.get_learned_conditioning(c_emb)
    uc = model.get_learned_conditioning(uc_emb)
    def step():
        samples, _ = sampler.sample(S=S, conditioning=c, batch_size=B, shape=[4, 64, 64], verbose=False,
                                    guidance_scale=[10.0, 4.0], unconditional_conditioning=uc, eta=0.0, x_T=x_T)
        frames = model.decode_first_stage_uint8(samples)
        return gather_frames(frames, global_batch=world * B)  # one RCCL all-gather per batch (no-op at N=1)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    timing = not args.no_kernel_timing
    if timing:
        lib.af_prof_reset()
        lib.af_prof_set_stride(args.event_stride)  # sample: an event pair per launch would cost ~10 % of the step
        lib.af_prof_enable(0b00011)  # conv_gemm + attention
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    lib.af_prof_enable(0)
    assert out.shape[-1] == 3 and out.dtype == torch.uint8
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    roof = None
    kernels = {}
    if timing:
        n = len(K_NAMES)
        ms = (C.c_double * n)()
        launches = (C.c_int64 * n)()
        flops = (C.c_double * n)()
        byts = (C.c_double * n)()
        _lib.check(lib.af_prof_collect(n, ms, launches, flops, byts), "af_prof_collect")
        for i, k in enumerate(K_NAMES):
            if launches[i]:
                kernels[k] = {"launches": int(launches[i]), "ms_total": ms[i], "avg_us": 1e3 * ms[i] / launches[i],
                              "algorithmic_tflops": flops[i] / (ms[i] * 1e-3) / 1e12 if flops[i] else None}
        if launches[0]:
            peak = PEAK_BF16 if args.dtype == "bf16" else PEAK_F32
            ach = flops[0] / (ms[0] * 1e-3)
            traffic = None
            tf = ROOT / "profiles" / "traffic_latest.json"
            if tf.exists():
                try:
                    traffic = json.loads(tf.read_text()).get("conv_gemm_hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            roof = {"kernel": "conv/linear class: conv_gemm_pp_kernel (256x160 ping-pong) + conv_gemm_kernel + splitk_reduce, all shapes",
                    "bound": "mfma", "achieved": ach / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
                    "frac": ach / peak, "traffic": traffic,
                    "flops_per_launch": flops[0] / launches[0], "avg_launch_us": 1e3 * ms[0] / launches[0],
                    "launches": int(launches[0]), "sampled_every": args.event_stride}

    if rank == 0:
        images = world * B * args.steps
        value = images / dt
        res = {
            "metric": "512x512 images/sec @ 50 DDIM steps, batch 8",
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"SD v1.5 512x512, {S} DDIM steps, batch {B} per GPU, CFG (Bf={2 * B}), "
                                   "layerwise 16x77x768 context, random-init weights", "global_batch": world * B,
                       "latent": [4, 64, 64], "guidance_scale": [10.0, 4.0], "parallelism": f"dp{world}"},
            "whole_path_algorithmic_tflops": value * (S * 2 * F_UNET + F_VAE) / 1e12 / world,
            "whole_path_frac_of_mfma_peak": value * (S * 2 * F_UNET + F_VAE) / world /
                                            (PEAK_BF16 if args.dtype == "bf16" else PEAK_F32),
            "roofline": roof, "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(S)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
