#!/usr/bin/env python
"""MI355X counterpart of the reference's scripts/stable_txt2img.py (the caller of the hot path).

Keeps the reference's flag names for everything that reaches the denoising path
(stable_txt2img.py:38-310): --config --ckpt --n_samples --n_repeat --ddim_steps --ddim_eta --scale --H --W --C --f
--seed --outdir --skip_save --fixed_code --gpu --bs --plms --init_img_paths --init_img_weight.  Text conditioning is the one difference: the CLIP tower /
EmbeddingManager are out of scope offline, so prompts are given as pre-computed embeddings
(--prompt_emb file.pt/.npy with a [B*16,77,768] or [77,768] tensor) or --synthetic.

    python scripts/stable_txt2img.py --synthetic --n_samples 8 --ddim_steps 50 --skip_save
    python scripts/stable_txt2img.py --synthetic --n_samples 64 --gpus 8      # starts the 8 ranks itself
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 scripts/stable_txt2img.py --synthetic --n_samples 64 --gpus 8
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, default=None, help="reference yaml (v1-inference-ada.yaml); default: built-in SD-1.5")
    ap.add_argument("--ckpt", type=str, default=None, help="SD checkpoint (.ckpt / .safetensors); default: seeded synthetic weights")
    ap.add_argument("--prompt_emb", type=str, default=None, help="pre-computed conditional embedding (.pt / .npy)")
    ap.add_argument("--neg_prompt_emb", type=str, default=None, help="pre-computed unconditional embedding")
    ap.add_argument("--synthetic", action="store_true", help="synthetic N(0,1) context of the reference's shape")
    ap.add_argument("--outdir", type=str, default="outputs/txt2img-samples")
    ap.add_argument("--skip_save", action="store_true", help="do not save individual samples (speed measurements)")
    ap.add_argument("--ddim_steps", type=int, default=50)
    ap.add_argument("--ddim_eta", type=float, default=0.0)
    ap.add_argument("--fixed_code", action="store_true", help="same starting code across repeats")
    ap.add_argument("--n_repeat", type=int, default=1)
    ap.add_argument("--n_samples", "--bs", dest="n_samples", type=int, default=8, help="global batch (sharded over ranks)")
    ap.add_argument("--H", type=int, default=512)
    ap.add_argument("--W", type=int, default=512)
    ap.add_argument("--C", type=int, default=4)
    ap.add_argument("--f", type=int, default=8)
    ap.add_argument("--scale", type=float, nargs="+", default=[10.0, 4.0], help="guidance scale, or max min for annealing")
    ap.add_argument("--init_img_paths", type=str, nargs="+", default=None,
                    help="initial image(s): encoded by the VAE encoder, averaged, blended with noise into the start code")
    ap.add_argument("--init_img_weight", type=float, default=0.1, help="w: start = w*enc(img) + (1-w)*noise")
    ap.add_argument("--plms", action="store_true", help="PLMS sampler instead of DDIM (scalar --scale)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--gpu", type=int, default=None)
    ap.add_argument("--gpus", type=int, default=1,
                    help="data-parallel ranks (one process per GPU); without a launcher the ranks are started here")
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    return ap.parse_args()


def load_img(path, h, w):
    """stable_txt2img.py:318-327: RGB, resized to multiples of 32, [-1, 1], NCHW."""
    from PIL import Image
    image = Image.open(path).convert("RGB")
    w, h = (x - x % 32 for x in (w, h))
    image = np.array(image.resize((w, h), resample=Image.LANCZOS)).astype(np.float32) / 255.0
    return 2.0 * torch.from_numpy(image[None].transpose(0, 3, 1, 2)) - 1.0


def load_emb(path, n, device):
    t = torch.tensor(np.load(path)) if path.endswith(".npy") else torch.load(path, map_location="cpu", weights_only=True)
    t = t.float()
    if t.dim() == 2:
        t = t[None].expand(n, -1, -1)
    if t.shape[0] == n:  # [B,77,768] -> layerwise [B*16,77,768] (embedding_manager.py:1342-1353)
        t = t[:, None].expand(n, 16, *t.shape[1:]).reshape(n * 16, *t.shape[1:])
    return t.contiguous().to(device)


def main():
    opt = parse_args()
    import torch.distributed as dist
    from adaface_amd.parallel import init_distributed, launch_ranks, launched_by_torchrun
    if opt.gpus > 1 and not launched_by_torchrun():   # before this process touches the GPU
        raise SystemExit(launch_ranks(opt.gpus, os.fspath(Path(__file__).resolve()), sys.argv[1:]))
    local = int(os.environ.get("LOCAL_RANK", "0")) if opt.gpu is None else opt.gpu
    if not torch.cuda.is_available():
        raise SystemExit("no HIP device: adaface_amd has no CPU path (use the reference for CPU plumbing runs)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    rank, world = init_distributed(opt.gpus, backend="nccl", device=device)   # fails if the group is not opt.gpus ranks
    from adaface_amd.configs import sd15_config
    from adaface_amd.parallel import gather_frames, shard_batch, shard_range
    from adaface_amd.synth import synth_context
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.util import instantiate_from_config, load_config, load_model_from_config

    config = load_config(opt.config) if opt.config else sd15_config()
    if opt.ckpt:
        model = load_model_from_config(config, opt.ckpt)
    else:
        model = instantiate_from_config(config["model"]).eval()
    model = model.to(device).set_compute_dtype(opt.dtype)
    if not opt.ckpt:  # seeded synthetic weights (identical on every rank)
        g = torch.Generator(device=device).manual_seed(1234)
        with torch.no_grad():
            for name, p in sorted(model.named_parameters()):
                if p.dim() == 1:
                    t = torch.randn(p.shape, generator=g, device=device)
                    p.copy_(1.0 + 0.1 * t if name.endswith(".weight") else 0.05 * t)
                else:
                    p.copy_(torch.randn(p.shape, generator=g, device=device) * p[0].numel() ** -0.5)
        model.model.diffusion_model._mark_dirty()
        model.first_stage_model._mark_dirty()

    B = opt.n_samples
    lo, hi = shard_range(B, rank, world)
    b = hi - lo
    if opt.prompt_emb:
        c_all = load_emb(opt.prompt_emb, B, "cpu")
        uc_all = load_emb(opt.neg_prompt_emb, B, "cpu") if opt.neg_prompt_emb else torch.zeros_like(c_all)
    elif opt.synthetic:
        c_all = synth_context(B, seed=opt.seed + 1, device="cpu")
        uc_all = synth_context(B, seed=opt.seed + 2, device="cpu", shared=True)
    else:
        raise SystemExit("give --prompt_emb (pre-computed CLIP/AdaFace embedding) or --synthetic")
    c = model.get_learned_conditioning(shard_batch(c_all, rank, world, per_sample=16).to(device))
    uc = model.get_learned_conditioning(shard_batch(uc_all, rank, world, per_sample=16).to(device))
    if opt.plms:
        from ldm.models.diffusion.plms import PLMSSampler
        sampler = PLMSSampler(model)
    else:
        sampler = DDIMSampler(model)
    shape = [opt.C, opt.H // opt.f, opt.W // opt.f]
    gs = opt.scale if len(opt.scale) > 1 else opt.scale[0]
    gen = torch.Generator().manual_seed(opt.seed)  # host RNG: the start code does not depend on the world size
    start_code = torch.randn([B] + shape, generator=gen) if opt.fixed_code else None
    if opt.init_img_paths:
        # stable_txt2img.py:594-625: encode each init image, average (divide by sqrt(N)), blend with noise
        avg = torch.zeros([B] + shape)
        for path in opt.init_img_paths:
            img = load_img(path, opt.H, opt.W).repeat(b, 1, 1, 1).to(device)
            enc = model.get_first_stage_encoding(model.encode_first_stage(img))   # this rank's samples
            avg[lo:hi] += enc.cpu()
        if world > 1:
            dist.all_reduce(avg_dev := avg.to(device))
            avg = avg_dev.cpu()
        avg /= np.sqrt(len(opt.init_img_paths))
        start_code = avg * opt.init_img_weight + torch.randn([B] + shape, generator=gen) * (1.0 - opt.init_img_weight)
    os.makedirs(opt.outdir, exist_ok=True)
    tic = time.time()
    count = 0
    with torch.no_grad(), model.ema_scope():
        for n in range(opt.n_repeat):
            x_T_all = start_code if start_code is not None else torch.randn([B] + shape, generator=gen)
            x_T = shard_batch(x_T_all, rank, world).to(device)
            if opt.plms:
                samples, _ = sampler.sample(S=opt.ddim_steps, conditioning=c, batch_size=b, shape=shape, verbose=False,
                                            unconditional_guidance_scale=opt.scale[0], unconditional_conditioning=uc,
                                            eta=opt.ddim_eta, x_T=x_T)
            else:
                samples, _ = sampler.sample(S=opt.ddim_steps, conditioning=c, batch_size=b, shape=shape, verbose=False,
                                            guidance_scale=gs, unconditional_conditioning=uc, eta=opt.ddim_eta, x_T=x_T)
            frames = gather_frames(model.decode_first_stage_uint8(samples), global_batch=B)
            if rank == 0 and not opt.skip_save:
                from PIL import Image
                for i, f in enumerate(frames.cpu().numpy()):
                    Image.fromarray(f).save(os.path.join(opt.outdir, f"{count:05}.jpg"))
                    count += 1
    torch.cuda.synchronize()
    toc = time.time()
    if rank == 0:
        n_img = B * opt.n_repeat
        print(f"{n_img} images of {opt.H}x{opt.W} @ {opt.ddim_steps} DDIM steps in {toc - tic:.2f} s "
              f"({n_img / (toc - tic):.2f} images/s incl. first-call warm-up) on {world} GPU(s); outputs: {opt.outdir}")
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
