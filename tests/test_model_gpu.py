"""GPU parity of the whole networks (through the C ABI: af_create / af_load_tensor / af_set_context /
af_unet_forward / af_vae_decode / af_ddim_step) against

  (1) the committed golden vectors produced by the reference's own modules (tests/golden/*.npz), and
  (2) the CPU oracle (oracle/ldm_oracle.py) on the same seeded weights and inputs.

Tolerances (relative to max|reference| of the compared tensor):
  f32 mode  (parity mode, f32 MFMA):  UNet eps / VAE image 2e-4;  5-step DDIM latent 1e-3 (north star: <= 1e-3)
  bf16 mode (throughput mode):        UNet eps 3e-2, VAE image 3e-2 (measured 1.1-1.8e-2) — the MEASURED bf16 deviation is
                                      written to gpurun_out/parity_report.txt; SURVEY.md §7.3-1: bf16 cannot meet 1e-3.
"""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import ldm_oracle as O  # noqa: E402  (the checker, never the thing measured)

pytestmark = pytest.mark.gpu
GOLD = ROOT / "tests" / "golden"
TOL = {"f32": 2e-4, "bf16": 3e-2}
TAP_TOL = {"f32": 2e-4, "bf16": 5e-2}   # 64 samples normalised by the block's own sample scale: noisier than a whole tensor


def _unet_kwargs(cfg: O.UNetConfig):
    return dict(in_channels=cfg.in_channels, model_channels=cfg.model_channels, out_channels=cfg.out_channels,
                num_res_blocks=cfg.num_res_blocks, attention_resolutions=cfg.attention_resolutions,
                channel_mult=cfg.channel_mult, num_heads=cfg.num_heads, context_dim=cfg.context_dim,
                transformer_depth=cfg.transformer_depth, n_context_layers=cfg.n_context_layers)


def _vae_kwargs(cfg: O.VAEConfig):
    return dict(ch=cfg.ch, out_ch=cfg.out_ch, ch_mult=cfg.ch_mult, num_res_blocks=cfg.num_res_blocks,
                z_channels=cfg.z_channels, embed_dim=cfg.embed_dim)


def _rel(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    if not np.isfinite(got).all():
        return float("inf")
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12))


def _check_taps(eng, x, t, gold, prefix, dtype, report):
    """Per-block outputs (forward hooks on input_blocks[i] / middle_block / output_blocks[j] of the REFERENCE UNet,
    tests/golden/gen_golden.py) against the HIP path's block outputs through the diagnostic tap: localises a deviation
    that the eps comparison would only show as a number.  Samples = 64 strided elements per block; the error is relative
    to the block's own scale (max |sample|, at least its std)."""
    n_in = sum(1 for k in gold if k.startswith(f"{prefix}_tap_input_blocks.") and k.endswith("_sample"))
    n_out = sum(1 for k in gold if k.startswith(f"{prefix}_tap_output_blocks.") and k.endswith("_sample"))
    names = [f"input_blocks.{i}" for i in range(n_in)] + ["middle_block"] + [f"output_blocks.{j}" for j in range(n_out)]
    outs = eng.unet_block_outputs(x, t)
    assert len(outs) == len(names)
    worst = (0.0, "")
    for b, name in enumerate(names):
        flat = outs[b].reshape(-1)
        sample = flat[:: max(1, flat.numel() // 64)][:64].cpu().numpy()
        ref = gold[f"{prefix}_tap_{name}_sample"]
        scale = max(float(np.abs(ref).max()), float(gold[f"{prefix}_tap_{name}_stats"][1]))
        err = float(np.abs(sample - ref).max() / scale)
        worst = max(worst, (err, name))
        assert np.isfinite(sample).all() and err < TAP_TOL[dtype], (name, err)
    report(f"{prefix}_unet worst block tap ({worst[1]}) vs reference golden [{dtype}]", worst[0], 1.0, TAP_TOL[dtype])


@pytest.fixture(scope="module")
def tiny():
    return dict(np.load(GOLD / "golden_tiny.npz"))


def test_param_inventory_agrees(gpu):
    """C library slots == product-side layout == oracle inventory (names and shapes)."""
    from adaface_amd import layout
    from adaface_amd.engine import Engine
    cfg = O.SD15_UNET
    eng = Engine(dtype="bf16", unet=_unet_kwargs(cfg), vae=_vae_kwargs(O.SD15_VAE))
    table = eng.tensor_table()
    ora = dict(O.unet_param_shapes(cfg))
    ora.update(O.vae_param_shapes(O.SD15_VAE))
    lay = {"model.diffusion_model." + k: v for k, v in layout.unet_param_shapes(**_unet_kwargs(cfg)).items()}
    lay.update({"first_stage_model.decoder." + k: v
                for k, v in layout.vae_decoder_param_shapes(**_vae_kwargs(O.SD15_VAE)).items()})
    lay["first_stage_model.post_quant_conv.weight"] = (4, 4, 1, 1)
    lay["first_stage_model.post_quant_conv.bias"] = (4,)
    assert set(table) == set(ora) == set(lay)
    for k, shp in ora.items():
        assert int(np.prod(table[k])) == int(np.prod(shp)), k
        assert tuple(lay[k]) == tuple(shp), k
    eng.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_tiny_unet(gpu, report, tiny, dtype):
    from adaface_amd.engine import Engine
    cfg = O.TINY_UNET
    sd = O.synth_state_dict(O.unet_param_shapes(cfg), seed=11)
    eng = Engine(dtype=dtype, unet=_unet_kwargs(cfg))
    assert eng.load_state_dict(sd) == []
    x = torch.tensor(tiny["tiny_x"], device=gpu)
    t = torch.tensor(tiny["tiny_t"], device=gpu)
    ctx = torch.tensor(tiny["tiny_ctx"], device=gpu)
    eng.set_context(ctx, x.shape[0], layerwise=True)
    eps = eng.unet_forward(x, t).cpu().numpy()
    err = _rel(eps, tiny["tiny_eps"])
    report(f"tiny_unet eps vs reference golden [{dtype}]", err, 1.0, TOL[dtype])
    assert err < TOL[dtype], err
    # second call with the cached context must be bit-identical (no stale state in the arena)
    eps2 = eng.unet_forward(x, t).cpu().numpy()
    assert np.array_equal(eps, eps2)
    _check_taps(eng, x, t, tiny, "tiny", dtype, report)
    eng.close()


def test_tiny_unet_plain_context(gpu, report):
    """use_layerwise_context=False path (openaimodel.py:871-872): one [B,T,D] context for all layers."""
    from adaface_amd.engine import Engine
    cfg = O.TINY_UNET
    sd = O.synth_state_dict(O.unet_param_shapes(cfg), seed=11)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(3, 4, 32, 16, generator=g)       # ragged batch / non-square latent
    t = torch.tensor([901, 401, 1])
    ctx = torch.randn(3, 77, cfg.context_dim, generator=g)
    ref = O.unet_forward(sd, cfg, x, t, ctx, use_layerwise_context=False).numpy()
    eng = Engine(dtype="f32", unet=_unet_kwargs(cfg))
    eng.load_state_dict(sd)
    eng.set_context(ctx.to(gpu), 3, layerwise=False)
    eps = eng.unet_forward(x.to(gpu), t.to(gpu)).cpu().numpy()
    err = _rel(eps, ref)
    report("tiny_unet plain ctx, B3 32x16 vs oracle [f32]", err, 1.0, TOL["f32"])
    assert err < TOL["f32"], err
    eng.close()


def test_tiny_ddim_trajectory(gpu, report, tiny):
    """5 DDIM steps (annealed guidance 10->4, CFG batch doubling) through the HIP UNet + fused update kernel,
    against the latent produced by the reference's DDIMSampler driving the reference UNet."""
    from adaface_amd import ops
    from adaface_amd.engine import Engine
    cfg = O.TINY_UNET
    sd = O.synth_state_dict(O.unet_param_shapes(cfg), seed=11)
    eng = Engine(dtype="f32", unet=_unet_kwargs(cfg))
    eng.load_state_dict(sd)
    S = 5
    sched = O.register_schedule()
    ts = O.make_ddim_timesteps(S)
    sig, a, ap = O.make_ddim_sampling_parameters(sched["alphas_cumprod"], ts, 0.0)
    gs = O.guidance_schedule((10.0, 4.0), S)
    img = torch.tensor(tiny["ddim_xT"], device=gpu)
    c = torch.tensor(tiny["ddim_c"], device=gpu)
    uc = torch.tensor(tiny["ddim_uc"], device=gpu)
    b = img.shape[0]
    eng.set_context(torch.cat([c, uc]), 2 * b, layerwise=True)
    for i, step in enumerate(np.flip(ts)):
        idx = S - i - 1
        t = torch.full((2 * b,), int(step), device=gpu, dtype=torch.long)
        e = eng.unet_forward(torch.cat([img, img]), t)
        img, _ = ops.ddim_step(img, e[:b], e[b:], gs[i], float(a[idx]), float(ap[idx]),
                               float(np.sqrt(1.0 - a[idx].item())), float(sig[idx]))
    err = _rel(img.cpu().numpy(), tiny["ddim_S5_samples"])
    report("tiny DDIM S=5 final latent vs reference sampler [f32]", err, float(np.abs(tiny["ddim_S5_samples"]).max()), 1e-3)
    assert err < 1e-3, err
    eng.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_tiny_vae(gpu, report, tiny, dtype):
    from adaface_amd.engine import Engine
    cfg = O.TINY_VAE
    sd = O.synth_state_dict(O.vae_param_shapes(cfg), seed=12)
    eng = Engine(dtype=dtype, vae=_vae_kwargs(cfg))
    eng.load_state_dict(sd)
    z = torch.tensor(tiny["vae_z"], device=gpu)
    img, u8 = eng.vae_decode(z, scale_factor=cfg.scale_factor, want_uint8=True)
    err = _rel(img.cpu().numpy(), tiny["vae_tiny_img"])
    report(f"tiny_vae image vs reference golden [{dtype}]", err, 1.0, TOL[dtype])
    assert err < TOL[dtype], err
    ref_u8 = O.to_uint8_hwc(torch.tensor(tiny["vae_tiny_img"]))
    d = np.abs(u8.cpu().numpy().astype(int) - ref_u8.astype(int))
    assert d.max() <= (1 if dtype == "f32" else 12), d.max()
    eng.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_sd15_unet_golden(gpu, report, dtype):
    """Full SD-1.5 UNet (859.5 M params) on one CFG pair at 64x64 vs the reference's eps and per-block samples."""
    from adaface_amd.engine import Engine
    g = dict(np.load(GOLD / "golden_sd15.npz"))
    cfg = O.SD15_UNET
    sd = O.synth_state_dict(O.unet_param_shapes(cfg), seed=21)
    eng = Engine(dtype=dtype, unet=_unet_kwargs(cfg))
    assert eng.load_state_dict(sd) == []
    del sd
    x = torch.tensor(g["sd15_x"], device=gpu)
    t = torch.tensor(g["sd15_t"], device=gpu)
    ctx = torch.tensor(g["sd15_ctx"]).float().to(gpu)
    eng.set_context(ctx, 2, layerwise=True)
    eps = eng.unet_forward(x, t).cpu().numpy()
    err = _rel(eps, g["sd15_eps"])
    report(f"sd15_unet eps vs reference golden [{dtype}]", err, float(np.abs(g["sd15_eps"]).max()), TOL[dtype])
    assert err < TOL[dtype], err
    _check_taps(eng, x, t, g, "sd15", dtype, report)
    eng.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_sd15_vae_golden(gpu, report, dtype):
    from adaface_amd.engine import Engine
    g = dict(np.load(GOLD / "golden_sd15.npz"))
    cfg = O.SD15_VAE
    sd = O.synth_state_dict(O.vae_param_shapes(cfg), seed=22)
    eng = Engine(dtype=dtype, vae=_vae_kwargs(cfg))
    eng.load_state_dict(sd)
    z = torch.tensor(g["sd15_vae_z"], device=gpu)
    img = eng.vae_decode(z, scale_factor=cfg.scale_factor).cpu().numpy()
    scale = float(g["sd15_vae_img_stats"][2])
    e1 = np.abs(img[:, :, 192:320, 192:320] - g["sd15_vae_img_crop"]).max() / scale
    e2 = np.abs(img[:, :, ::8, ::8] - g["sd15_vae_img_sub8"]).max() / scale
    err = float(max(e1, e2))
    report(f"sd15_vae image vs reference golden [{dtype}]", err, scale, TOL[dtype])
    assert np.isfinite(img).all()
    assert err < TOL[dtype], err
    eng.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_sd15_vae_encoder_golden(gpu, report, dtype):
    """Full-size VAE encoder (512x512 image -> posterior moments [1,8,64,64]) vs the reference Encoder + quant_conv."""
    from adaface_amd.engine import Engine
    g = dict(np.load(GOLD / "golden_sd15.npz"))
    cfg = O.SD15_VAE
    sd = O.synth_state_dict(O.vae_encoder_param_shapes(cfg), seed=23)
    eng = Engine(dtype=dtype, vae=dict(_vae_kwargs(cfg), encoder=True, in_channels=3))
    eng.load_state_dict(sd, strict=False)   # encoder side only
    x = torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(int(g["sd15_enc_x_seed"][0]))) * 2.0 - 1.0
    mom = eng.vae_encode(x.to(gpu)).cpu().numpy()
    ref = g["sd15_enc_moments"]
    err = _rel(mom, ref)
    report(f"sd15_vae encoder moments vs reference golden [{dtype}]", err, float(np.abs(ref).max()), TOL[dtype])
    assert mom.shape == ref.shape and err < TOL[dtype], err
    eng.close()


# ---------------------------------------------------------------------------------------------------------------
# Parity at the BENCHMARK's shape.  The planner (af_plan_conv_gemm) keys on grid fill, so at Bf = 16 (M = 65536 /
# 16384 / 4096 / 1024) it picks other tiles, split-K factors and tile orders than at Bf = 2; the launches bench.py
# times are reached here through the whole model and compared with the same samples run as Bf = 2 pairs (whose path
# is pinned against the reference goldens above).  Samples never interact, so the two must agree up to the summation
# order of the chosen tilings:  f32 mode <= 1e-5 of max|eps|.
#
# bf16 mode: the bars are DERIVED inside each test, not tuned.  Every bf16 A/B (two tilings, fused vs stand-alone
# LayerNorm, twin vs concatenated CFG batch) also runs the f32-mode forward of the same batch and asserts
#   (1) each bf16 forward is within BF16_FWD_BAR of the f32-mode forward: max-abs / max|eps| <= 3e-2, the package's stated
#       per-forward bf16 bar since round 1 (TOL["bf16"]).  The maximum over 10^5-10^6 elements is an extreme-value
#       statistic: across batches, tilings and kernel variants it has measured 1.2e-2 ... 2.0e-2 (a 2e-2 bar tried in
#       round 3 sat inside that spread: 2.02e-2 on the Bf = 2 twin forward), so the drift-sensitive assertions are the RMS
#       ones: rms(error) / rms(eps) <= BF16_FWD_RMS_BAR = 2e-2 (measured 1.19e-2 ... 1.62e-2 depending on the inputs), and
#       the two forwards of an A/B must have the SAME rms error against the f32 mode to within 10 % (measured: they agree
#       to 1 % -- 1.546e-2 vs 1.545e-2, 1.616e-2 vs 1.603e-2: a variant that only rounds differently does not move the
#       rms; one whose drift grows does), and
#   (2) the two bf16 forwards differ by at most AB_MARGIN * sqrt(2) * max(e_a, e_b): two forwards whose roundings are
#       independent are sqrt(2) * e apart in the max norm, 25 % margin for the max statistics of two finite samples
#       (the triangle inequality alone would allow 2 * e).  A fused path whose drift doubles fails (1).
# ---------------------------------------------------------------------------------------------------------------
BATCH_TOL = {"f32": 1e-5}
BF16_FWD_BAR = 3e-2
BF16_FWD_RMS_BAR = 2e-2
AB_MARGIN = 1.25


def _rms_rel(a, ref):
    return ((a - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()


def _f32_mode_forward(gpu, cfg, seed, x, t, ctx, Bf):
    """The f32 (parity) mode forward of a batch: the reference every bf16 A/B below is measured against."""
    from adaface_amd.engine import Engine
    from adaface_amd.synth import synth_weights_into
    eng = Engine(dtype="f32", unet=_unet_kwargs(cfg))
    synth_weights_into(eng, O.unet_param_shapes(cfg), seed=seed, device=gpu)
    eng.set_context(ctx, Bf, layerwise=True)
    out = eng.unet_forward(x, t)
    eng.close()
    return out


def _assert_bf16_ab(report, name, a, b, ref):
    """Derived bf16 bars (see the block comment above): a, b = two bf16 forwards, ref = the f32-mode forward."""
    scale = ref.abs().max().item()
    e_a = (a - ref).abs().max().item() / scale
    e_b = (b - ref).abs().max().item() / scale
    d = (a - b).abs().max().item() / scale
    bar = AB_MARGIN * 2.0 ** 0.5 * max(e_a, e_b)
    r_a, r_b = _rms_rel(a, ref), _rms_rel(b, ref)
    report(f"{name}: bf16 forward A vs f32 mode", e_a, scale, BF16_FWD_BAR)
    report(f"{name}: bf16 forward B vs f32 mode", e_b, scale, BF16_FWD_BAR)
    report(f"{name}: bf16 forward A vs f32 mode, rms / rms", r_a, 1.0, BF16_FWD_RMS_BAR)
    report(f"{name}: bf16 forward B vs f32 mode, rms / rms", r_b, 1.0, BF16_FWD_RMS_BAR)
    report(f"{name}: A vs B (bar = 1.25 * sqrt(2) * measured per-forward error)", d, scale, bar)
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    assert e_a <= BF16_FWD_BAR and e_b <= BF16_FWD_BAR, (name, e_a, e_b)
    assert r_a <= BF16_FWD_RMS_BAR and r_b <= BF16_FWD_RMS_BAR, (name, r_a, r_b)
    assert abs(r_a - r_b) <= 0.1 * max(r_a, r_b), (name, r_a, r_b)
    assert d <= bar, (name, d, e_a, e_b)
    return e_a, e_b, d


def test_sd15_unet_layernorm_folding_ab(gpu, report, knobs):
    """bf16 at the benchmark batch: the LayerNorm-folded transformer GEMMs (mu / rstd applied in the consumer's epilogue on
    W * gamma, row statistics from the producer's epilogue) against the same forward with stand-alone LayerNorm kernels.
    Both are bf16 forwards that round differently: each is held to the per-forward bar against the f32-mode forward and
    the pair to sqrt(2) x the measured per-forward error (_assert_bf16_ab)."""
    from adaface_amd import _lib
    from adaface_amd.engine import Engine
    from adaface_amd.synth import synth_weights_into
    cfg = O.SD15_UNET
    g = torch.Generator().manual_seed(35)
    x = torch.randn(16, 4, 64, 64, generator=g).to(gpu)
    t = torch.full((16,), 301, dtype=torch.long, device=gpu)
    ctx = torch.randn(16 * 16, 77, cfg.context_dim, generator=g).to(gpu)
    eng = Engine(dtype="bf16", unet=_unet_kwargs(cfg))
    synth_weights_into(eng, O.unet_param_shapes(cfg), seed=36, device=gpu)
    eng.set_context(ctx, 16, layerwise=True)
    _lib.plan_counts(reset=True)
    eps_fused = eng.unet_forward(x, t)
    pc = _lib.plan_counts(reset=True)
    knobs("ln_fuse", 0)
    eps_plain = eng.unet_forward(x, t)
    pc0 = _lib.plan_counts(reset=True)
    assert pc["ln_consumer"] == 25 and pc["xattn_fused"] == 5 and pc0["ln_consumer"] == 0 and pc0["ln_producer"] == 0 and pc0["xattn_fused"] == 0, (pc, pc0)
    eng.close()
    ref = _f32_mode_forward(gpu, cfg, 36, x, t, ctx, 16)
    _assert_bf16_ab(report, "sd15_unet Bf=16 LayerNorm-folded (A) vs stand-alone LayerNorm (B)", eps_fused, eps_plain, ref)


def test_sd15_unet_new_fused_paths_ab(gpu, report, knobs):
    """Round-3 / round-4 fusions at the benchmark batch, each against the forward without it (knob off) and both against the
    f32-mode forward (derived bf16 bars, _assert_bf16_ab, unchanged): the cross-attention layer of the 64x64 level as one
    kernel (xattn_fused_kernel: to_q + attention + to_out + residual), the register-resident short-key cross-attention
    kernel, and the SpatialTransformer GroupNorm applied in the prologue of the row-panel proj_in."""
    from adaface_amd import _lib
    from adaface_amd.engine import Engine
    from adaface_amd.synth import synth_weights_into
    cfg = O.SD15_UNET
    g = torch.Generator().manual_seed(61)
    x = torch.randn(16, 4, 64, 64, generator=g).to(gpu)
    t = torch.randint(0, 1000, (16,), generator=g).to(gpu)
    ctx = torch.randn(16 * 16, 77, cfg.context_dim, generator=g).to(gpu)
    eng = Engine(dtype="bf16", unet=_unet_kwargs(cfg))
    synth_weights_into(eng, O.unet_param_shapes(cfg), seed=62, device=gpu)
    eng.set_context(ctx, 16, layerwise=True)
    _lib.plan_counts(reset=True)
    full = eng.unet_forward(x, t)
    pc = _lib.plan_counts(reset=True)
    # (the five cross-attention layers of the 64x64 level run as ONE kernel each, round 4: the short-key kernel keeps the 32x32 level)
    assert pc["attn_short"] == 5 and pc["gn_consumer"] == 5 and pc["xattn_fused"] == 5, pc
    outs = {}
    for knob in ("xattn_fused", "attn_short", "gn_consumer"):
        knobs(knob, 0)
        if knob == "attn_short":
            knobs("xattn_fused", 0)           # (all ten short-key launches on the flash kernel: the round-3 A/B)
        outs[knob] = eng.unet_forward(x, t)
        pc0 = _lib.plan_counts(reset=True)
        assert pc0[knob] == 0, pc0
        if knob == "xattn_fused":
            assert pc0["attn_short"] == 10, pc0   # three launches per layer again
        knobs(knob, 1)
        knobs("xattn_fused", 1)
    eng.close()
    ref = _f32_mode_forward(gpu, cfg, 62, x, t, ctx, 16)
    for knob, out in outs.items():
        _assert_bf16_ab(report, f"sd15_unet Bf=16 with (A) / without (B) {knob}", full, out, ref)


@pytest.mark.parametrize("B,H,W", [(16, 48, 48), (6, 96, 64), (3, 40, 24), (6, 64, 64), (10, 64, 64), (14, 64, 64)])
def test_sd15_unet_other_latent_sizes(gpu, report, B, H, W):
    """Latent sizes other than the benchmark's 64 x 64 (the planner's fall-backs: 48 x 48 -> 2304 pixels per sample, row-panel
    launches and the GroupNorm-in-prologue path but no LDS-halo convolution (Wo = 48); 96 x 64 -> halo rows of a non
    power-of-two image height; 40 x 24 with an odd batch -> below every row-panel threshold, ragged query blocks in the
    short-key attention) and 64 x 64 with CFG batches of 3 / 5 / 7 images (384 / 640 / 896 rows on the 8 x 8 map: ragged row
    tiles of the sliced-K launches, 1536 / 2560 / 3584 rows at 16 x 16: the 128 x 160 GEMM off and on its row limits): the bf16
    forward against the f32-mode forward of the same batch, per-forward bars."""
    from adaface_amd.engine import Engine
    from adaface_amd.synth import synth_weights_into
    cfg = O.SD15_UNET
    g = torch.Generator().manual_seed(H * W + B)
    x = torch.randn(B, 4, H, W, generator=g).to(gpu)
    t = torch.randint(0, 1000, (B,), generator=g).to(gpu)
    ctx = torch.randn(B * 16, 77, cfg.context_dim, generator=g).to(gpu)
    out = {}
    for dtype in ("f32", "bf16"):
        eng = Engine(dtype=dtype, unet=_unet_kwargs(cfg))
        synth_weights_into(eng, O.unet_param_shapes(cfg), seed=71, device=gpu)
        eng.set_context(ctx, B, layerwise=True)
        out[dtype] = eng.unet_forward(x, t)
        again = eng.unet_forward(x, t)
        assert torch.equal(out[dtype], again)                 # run-to-run determinism at this shape
        eng.close()
    scale = out["f32"].abs().max().item()
    e = (out["bf16"] - out["f32"]).abs().max().item() / scale
    r = _rms_rel(out["bf16"], out["f32"])
    report(f"sd15_unet B={B} {H}x{W} bf16 forward vs f32 mode", e, scale, BF16_FWD_BAR)
    report(f"sd15_unet B={B} {H}x{W} bf16 forward vs f32 mode, rms / rms", r, 1.0, BF16_FWD_RMS_BAR)
    assert torch.isfinite(out["bf16"]).all() and e <= BF16_FWD_BAR and r <= BF16_FWD_RMS_BAR, (e, r)


def test_sd15_unet_batch_consistency(gpu, report):
    """f32 mode: Bf = 16 vs Bf = 2 pairs agree to 1e-5 (summation order only).  bf16 mode: two tilings round their
    activations differently; the Bf = 16 forward and every Bf = 2 pair are held to the per-forward bar against the
    f32-mode forward of the same batch, and to sqrt(2) x the measured per-forward error against each other."""
    from adaface_amd import _lib
    from adaface_amd.engine import Engine
    from adaface_amd.synth import synth_weights_into
    cfg = O.SD15_UNET
    g = torch.Generator().manual_seed(32)
    B = 8                                    # images; CFG batch Bf = 16 = bench.py's forward
    x = torch.randn(B, 4, 64, 64, generator=g)
    x = torch.cat([x, x]).to(gpu)            # cat([x] * 2) as p_sample_ddim (ddim.py:233)
    t = torch.full((2 * B,), 701, dtype=torch.long, device=gpu)
    ctx = torch.randn(2 * B * 16, 77, cfg.context_dim, generator=g).to(gpu)   # per-layer different context rows
    eps_f32 = None
    for dtype in ("f32", "bf16"):
        eng = Engine(dtype=dtype, unet=_unet_kwargs(cfg))
        synth_weights_into(eng, O.unet_param_shapes(cfg), seed=31, device=gpu)
        assert eng.missing_tensors() == []
        _lib.plan_counts(reset=True)
        eng.set_context(ctx, 2 * B, layerwise=True)
        eps16 = eng.unet_forward(x, t)
        pc = _lib.plan_counts(reset=True)
        if dtype == "bf16":      # the launches the benchmark times: eight-wave ping-pong tiles and a sliced-K launch,
            # and the LayerNorm-folded GEMMs of the 64x64 and 32x32 transformers (3 consumers + 3 producers each)
            assert pc["tile4"] > 0 and pc["tile5"] > 0 and pc["splitk"] > 0, pc
            # (round 4: the 64x64 level's attn2.to_q consumer and attn2.to_out producer live inside xattn_fused_kernel: 30 - 5 each)
            assert pc["ln_consumer"] == 25 and pc["ln_producer"] == 25, pc
            # the three Upsample convolutions as four 2x2 phase convolutions; row-panel GEMMs at the 64x64 and 32x32 levels
            assert pc["up_phase4"] == 3 and pc["rowpanel"] >= 25, pc
            # ResBlock convolutions at the 64x64 / 32x32 levels that also summed the GroupNorm statistics of their output
            assert pc["gn_producer"] >= 15, pc
            # the five cross-attention layers of the 64x64 level as ONE kernel each, those of the 32x32 level on the
            # register-resident short-key kernel
            assert pc["xattn_fused"] == 5 and pc["attn_short"] == 5, pc
            # ... and the GroupNorm of the 64x64-level ones applied in the prologue of the row-panel proj_in
            assert pc["gn_consumer"] == 5, pc
            e16 = (eps16 - eps_f32).abs().max().item() / eps_f32.abs().max().item()
            report("sd15_unet Bf=16 bf16 forward vs f32-mode forward of the same batch", e16, eps_f32.abs().max().item(), BF16_FWD_BAR)
            assert e16 <= BF16_FWD_BAR, e16
        else:                    # parity mode: four-wave tiles, LDS-halo 3x3 kernel, sliced K
            assert pc["halo"] > 0 and pc["splitk"] > 0 and pc["tile4"] == 0 and pc["tile5"] == 0, pc
            eps_f32 = eps16
        scale = eps_f32.abs().max().item()
        worst, worst_fwd = 0.0, 0.0
        for b in range(B):       # sample b as its own CFG pair (cond b, uncond b)
            idx = torch.tensor([b, B + b], device=gpu)
            rows = torch.cat([torch.arange(16 * b, 16 * b + 16), torch.arange(16 * (B + b), 16 * (B + b) + 16)]).to(gpu)
            eng.set_context(ctx[rows].contiguous(), 2, layerwise=True)
            eps2 = eng.unet_forward(x[idx].contiguous(), t[idx].contiguous())
            assert torch.isfinite(eps2).all()
            worst = max(worst, (eps2 - eps16[idx]).abs().max().item() / scale)
            if dtype == "bf16":
                worst_fwd = max(worst_fwd, (eps2 - eps_f32[idx]).abs().max().item() / scale)
        if dtype == "f32":
            report("sd15_unet Bf=16 vs the same samples as Bf=2 pairs [f32]", worst, scale, BATCH_TOL["f32"])
            assert worst <= BATCH_TOL["f32"], (worst, pc)
        else:
            bar = AB_MARGIN * 2.0 ** 0.5 * max(e16, worst_fwd)
            report("sd15_unet Bf=2 pairs, bf16 forward vs f32-mode forward", worst_fwd, scale, BF16_FWD_BAR)
            report("sd15_unet Bf=16 vs the same samples as Bf=2 pairs [bf16] (bar = 1.25 * sqrt(2) * measured per-forward error)",
                   worst, scale, bar)
            assert worst_fwd <= BF16_FWD_BAR, worst_fwd
            assert worst <= bar, (worst, e16, worst_fwd, pc)
        eng.close()


@pytest.mark.parametrize("dtype,B", [("f32", 2), ("bf16", 8), ("bf16", 1)])
def test_sd15_unet_forward_twin(gpu, report, dtype, B):
    """af_unet_forward_twin(x, t) == af_unet_forward(cat([x] * 2), cat([t] * 2)): the classifier-free-guidance batch with
    its context-independent prefix (time embedding, conv_in, the first ResBlock, the first transformer up to the
    cross-attention) computed once and copied.  f32: equal up to summation order (1e-5); bf16: the half-batch launches of
    the prefix may be planned differently (other tile / K slicing), so the two forwards are held to the derived bf16 bars
    (_assert_bf16_ab) -- and the two halves of the twin forward must differ (they see different contexts)."""
    from adaface_amd.engine import Engine
    from adaface_amd.synth import synth_weights_into
    cfg = O.SD15_UNET
    g = torch.Generator().manual_seed(41 + B)
    x = torch.randn(B, 4, 64, 64, generator=g).to(gpu)
    t = torch.randint(0, 1000, (B,), generator=g).to(gpu)            # distinct timesteps per sample
    ctx = torch.randn(2 * B * 16, 77, cfg.context_dim, generator=g).to(gpu)
    eng = Engine(dtype=dtype, unet=_unet_kwargs(cfg))
    synth_weights_into(eng, O.unet_param_shapes(cfg), seed=42, device=gpu)
    eng.set_context(ctx, 2 * B, layerwise=True)
    full = eng.unet_forward(torch.cat([x, x]), torch.cat([t, t]))
    twin = eng.unet_forward_twin(x, t)
    assert twin.shape == full.shape and torch.isfinite(twin).all()
    scale = full.abs().max().item()
    if dtype == "f32":
        err = (twin - full).abs().max().item() / scale
        report(f"sd15_unet forward_twin vs forward(cat) B={B} [f32]", err, scale, BATCH_TOL["f32"])
        assert err <= BATCH_TOL["f32"], err
    else:
        ref = _f32_mode_forward(gpu, cfg, 42, torch.cat([x, x]), torch.cat([t, t]), ctx, 2 * B)
        _assert_bf16_ab(report, f"sd15_unet forward_twin (A) vs forward(cat) (B) B={B}", twin, full, ref)
    assert (twin[:B] - twin[B:]).abs().max().item() > 1e-3 * scale
    from adaface_amd._lib import AfError, check, ptr, stream_ptr
    with pytest.raises(AfError):                                      # an odd batch is not [x; x]
        out = torch.empty(3, 4, 64, 64, device=gpu)
        check(eng._lib.af_unet_forward_twin(eng._h, ptr(x), ptr(t), ptr(out), 3, 64, 64, stream_ptr()), "twin")
    eng.close()


def test_sd15_vae_batch_consistency(gpu, report):
    """VAE decode of the benchmark's 8 latents in one call vs each latent alone (float image and uint8 frame).  f32: equal to
    1e-5.  bf16: each decode within the per-forward bar of the f32-mode image, B = 8 vs B = 1 within sqrt(2) x that."""
    from adaface_amd import _lib
    from adaface_amd.engine import Engine
    from adaface_amd.synth import synth_weights_into
    cfg = O.SD15_VAE
    z = (torch.randn(8, 4, 64, 64, generator=torch.Generator().manual_seed(34)) * cfg.scale_factor).to(gpu)
    img_f32 = None
    for dtype in ("f32", "bf16"):
        eng = Engine(dtype=dtype, vae=_vae_kwargs(cfg))
        synth_weights_into(eng, O.vae_param_shapes(cfg), seed=33, device=gpu)
        _lib.plan_counts(reset=True)
        img8, u8 = eng.vae_decode(z, scale_factor=cfg.scale_factor, want_uint8=True)
        pc = _lib.plan_counts(reset=True)
        if dtype == "bf16":
            assert pc["tile4"] > 0, pc           # VAE widths (128 / 256 / 512) run on the 256x128 ping-pong tile
        else:
            img_f32 = img8
        scale = img_f32.abs().max().item()
        worst, worst_u8, worst_fwd = 0.0, 0, 0.0
        for b in (0, 3, 7):
            img1, u1 = eng.vae_decode(z[b:b + 1].contiguous(), scale_factor=cfg.scale_factor, want_uint8=True)
            worst = max(worst, (img1[0] - img8[b]).abs().max().item() / scale)
            worst_fwd = max(worst_fwd, (img1[0] - img_f32[b]).abs().max().item() / scale)
            worst_u8 = max(worst_u8, int((u1[0].int() - u8[b].int()).abs().max().item()))
        assert torch.isfinite(img8).all()
        if dtype == "f32":
            report("sd15_vae B=8 vs B=1 [f32]", worst, scale, BATCH_TOL["f32"])
            assert worst <= BATCH_TOL["f32"], worst
        else:
            e8 = (img8 - img_f32).abs().max().item() / scale
            bar = AB_MARGIN * 2.0 ** 0.5 * max(e8, worst_fwd)
            report("sd15_vae B=8 bf16 decode vs f32-mode decode", e8, scale, BF16_FWD_BAR)
            report("sd15_vae B=8 vs B=1 [bf16] (bar = 1.25 * sqrt(2) * measured per-decode error)", worst, scale, bar)
            assert e8 <= BF16_FWD_BAR and worst_fwd <= BF16_FWD_BAR, (e8, worst_fwd)
            assert worst <= bar, (worst, e8, worst_fwd)
        # bytes follow the floats: a float difference d moves 255 * (x + 1) / 2 by 127.5 d, plus one for a crossed boundary
        assert worst_u8 <= int(np.ceil(worst * scale * 127.5)) + 1, (worst_u8, worst * scale)
        eng.close()


def test_config0_sd15_256px_10steps_f32(gpu, report):
    """BASELINE.json configs[0] on the product path: SD v1.5 width, 256x256 (32x32 latent), 10 DDIM steps, batch 1,
    float32 — through the drop-in LatentDiffusion / DDIMSampler classes in f32 (parity) mode, against the CPU oracle's
    ddim_sample (reference ddim.py:135-220) and VAE decode on the same seeded weights.  North-star bar: final latent
    <= 1e-3 max-abs.  The product has no CPU path, so the f32-mode GPU run is this configuration's stand-in."""
    from adaface_amd.configs import sd15_config
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.util import instantiate_from_config
    model = instantiate_from_config(sd15_config()["model"]).eval()
    sd = O.synth_state_dict(O.unet_param_shapes(O.SD15_UNET), seed=21)
    sd.update(O.synth_state_dict(O.vae_param_shapes(O.SD15_VAE), seed=22))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("first_stage_model.encoder") or k.startswith("first_stage_model.quant_conv")
                                  or not k.startswith(("model.", "first_stage_model.")) for k in missing), missing[:5]
    model = model.to(gpu).set_compute_dtype("f32")
    g = torch.Generator().manual_seed(42)     # stable_txt2img.py:180 default seed
    S, B = 10, 1
    x_T = torch.randn(B, 4, 32, 32, generator=g)
    c = torch.randn(B * 16, 77, 768, generator=g)
    uc = torch.randn(B * 16, 77, 768, generator=g)
    sampler = DDIMSampler(model)
    assert list(np.flip(sampler_timesteps(S))) [:2] == [901, 801]
    samples, _ = sampler.sample(S=S, conditioning=model.get_learned_conditioning(c.to(gpu)), batch_size=B,
                                shape=[4, 32, 32], verbose=False, guidance_scale=[10.0, 4.0],
                                unconditional_conditioning=model.get_learned_conditioning(uc.to(gpu)), eta=0.0,
                                x_T=x_T.to(gpu))
    img = model.decode_first_stage(samples)
    u8 = model.decode_first_stage_uint8(samples)
    torch.cuda.synchronize()
    torch.set_num_threads(min(16, torch.get_num_threads() or 16))
    ref_lat = O.ddim_sample(lambda x, t, cc: O.unet_forward(sd, O.SD15_UNET, x, t, cc), O.register_schedule(), S, x_T, c, uc,
                            guidance_scale=(10.0, 4.0))
    ref_img = O.vae_decode(sd, O.SD15_VAE, ref_lat)
    e_abs = (samples.cpu() - ref_lat).abs().max().item()
    e_rel = e_abs / ref_lat.abs().max().item()
    e_img = (img.cpu() - ref_img).abs().max().item() / ref_img.abs().max().item()
    report("config0 SD-1.5 256px S=10 B=1 final latent MAX-ABS vs oracle [f32]", e_abs, ref_lat.abs().max().item(), 1e-3)
    report("config0 decoded image vs oracle [f32]", e_img, ref_img.abs().max().item(), 2e-4)
    assert e_abs <= 1e-3 and e_rel <= 1e-4, (e_abs, e_rel)
    assert e_img <= 2e-4, e_img
    ref_u8 = O.to_uint8_hwc(ref_img)
    d = np.abs(u8.cpu().numpy().astype(int) - ref_u8.astype(int))
    assert tuple(u8.shape) == (B, 256, 256, 3) and d.max() <= 1 and (d != 0).mean() < 2e-3, (d.max(), (d != 0).mean())


def sampler_timesteps(S):
    return O.make_ddim_timesteps(S)
