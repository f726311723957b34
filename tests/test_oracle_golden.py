"""Pin the CPU oracle (oracle/ldm_oracle.py) against golden vectors produced by the REFERENCE's
own modules (tests/golden/gen_golden.py; fixtures golden_tiny.npz / golden_sd15.npz).

fp32 CPU vs fp32 CPU of the same torch build: the restatement must agree to rounding —
tolerance 2e-5 absolute on O(1) activations (op order inside einsum/bmm may differ).
"""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import ldm_oracle as O  # noqa: E402

GOLD = ROOT / "tests" / "golden"


@pytest.fixture(scope="module")
def tiny():
    return dict(np.load(GOLD / "golden_tiny.npz"))


def test_schedule_tables(tiny):
    betas = O.make_beta_schedule()
    np.testing.assert_array_equal(betas, tiny["sched_betas"])
    np.testing.assert_array_equal(np.cumprod(1.0 - betas), tiny["sched_alphas_cumprod"])
    sched = O.register_schedule()
    for S in (10, 50):
        ts = O.make_ddim_timesteps(S)
        np.testing.assert_array_equal(ts, tiny[f"ddim_timesteps_S{S}"])
        sig, a, ap = O.make_ddim_sampling_parameters(sched["alphas_cumprod"], ts, 0.0)
        np.testing.assert_array_equal(a.numpy(), tiny[f"ddim_alphas_S{S}"])
        np.testing.assert_array_equal(ap, tiny[f"ddim_alphas_prev_S{S}"])
        np.testing.assert_array_equal(sig, tiny[f"ddim_sigmas_S{S}"])
    assert list(O.make_ddim_timesteps(50)[:3]) == [1, 21, 41] and O.make_ddim_timesteps(50)[-1] == 981
    assert list(O.make_ddim_timesteps(10)) == [1, 101, 201, 301, 401, 501, 601, 701, 801, 901]


def test_guidance_annealing():
    gs = O.guidance_schedule((10.0, 4.0), 50)
    assert gs[0] == 10.0 and abs(gs[-1] - 4.0) < 1e-9 and abs(gs[1] - 9.877551020408163) < 1e-12


def test_timestep_embedding(tiny):
    got = O.timestep_embedding(torch.tensor(tiny["temb_t"]), 320).numpy()
    np.testing.assert_array_equal(got, tiny["temb_320"])


def test_param_inventory_counts():
    n_unet = sum(int(np.prod(s)) for s in O.unet_param_shapes(O.SD15_UNET).values())
    n_vae = sum(int(np.prod(s)) for s in O.vae_param_shapes(O.SD15_VAE).values())
    assert n_unet == 859_520_964
    assert len(O.unet_param_shapes(O.SD15_UNET)) == 686
    # decoder 49,490,179 + post_quant_conv 4*4+4
    assert n_vae == 49_490_179 + 20


def test_tiny_unet_matches_reference(tiny):
    sd = O.synth_state_dict(O.unet_param_shapes(O.TINY_UNET), seed=11)
    taps = {}
    eps = O.unet_forward(sd, O.TINY_UNET, torch.tensor(tiny["tiny_x"]), torch.tensor(tiny["tiny_t"]),
                         torch.tensor(tiny["tiny_ctx"]), taps=taps)
    err = np.abs(eps.numpy() - tiny["tiny_eps"]).max()
    assert err < 2e-5, err
    assert np.abs(tiny["tiny_eps"]).max() > 0.05  # non-vacuous (zero-init tensors were re-randomised)
    for k, v in taps.items():
        flat = v.reshape(-1)
        sample = flat[:: max(1, flat.numel() // 64)][:64].numpy()
        ref = tiny[f"tiny_tap_{k}_sample"]
        assert np.abs(sample - ref).max() < 2e-5 * max(1.0, np.abs(ref).max()), k


def test_tiny_ddim_matches_reference_sampler(tiny):
    cfg = O.TINY_UNET
    sd = O.synth_state_dict(O.unet_param_shapes(cfg), seed=11)
    apply = lambda x, t, c: O.unet_forward(sd, cfg, x, t, c)
    out = O.ddim_sample(apply, O.register_schedule(), 5, torch.tensor(tiny["ddim_xT"]), torch.tensor(tiny["ddim_c"]),
                        torch.tensor(tiny["ddim_uc"]), guidance_scale=(10.0, 4.0))
    ref = tiny["ddim_S5_samples"]  # guidance 10 on random weights: |x| reaches ~30
    err = np.abs(out.numpy() - ref).max()
    assert err < 1e-5 * np.abs(ref).max(), err


def test_tiny_plms_matches_reference_sampler(tiny):
    cfg = O.TINY_UNET
    sd = O.synth_state_dict(O.unet_param_shapes(cfg), seed=11)
    apply = lambda x, t, c: O.unet_forward(sd, cfg, x, t, c)
    out = O.plms_sample(apply, O.register_schedule(), 6, torch.tensor(tiny["ddim_xT"]), torch.tensor(tiny["ddim_c"]),
                        torch.tensor(tiny["ddim_uc"]), guidance_scale=3.0)
    ref = tiny["plms_S6_samples"]
    assert np.abs(out.numpy() - ref).max() < 1e-5 * np.abs(ref).max()
    # S=6 -> 7 actual steps (1000 // 6 = 166): annealing and table indices run over len(timesteps), not S
    out = O.ddim_sample(apply, O.register_schedule(), 6, torch.tensor(tiny["ddim_xT"]), torch.tensor(tiny["ddim_c"]),
                        torch.tensor(tiny["ddim_uc"]), guidance_scale=(6.0, 2.0))
    ref = tiny["ddim_S6_samples"]
    assert np.abs(out.numpy() - ref).max() < 1e-5 * np.abs(ref).max()


def test_tiny_vae_matches_reference(tiny):
    sd = O.synth_state_dict(O.vae_param_shapes(O.TINY_VAE), seed=12)
    img = O.vae_decode(sd, O.TINY_VAE, torch.tensor(tiny["vae_z"]))
    ref = tiny["vae_tiny_img"]
    assert img.shape == ref.shape
    err = np.abs(img.numpy() - ref).max()
    assert err < 2e-5 * max(1.0, np.abs(ref).max()), err


@pytest.mark.skipif(not (GOLD / "golden_sd15.npz").exists(), reason="full-size fixture not generated")
def test_sd15_unet_matches_reference():
    """Full SD-1.5 UNet (859.5 M params), one CFG pair at 64x64: ~25 s of CPU."""
    g = dict(np.load(GOLD / "golden_sd15.npz"))
    sd = O.synth_state_dict(O.unet_param_shapes(O.SD15_UNET), seed=21)
    taps = {}
    eps = O.unet_forward(sd, O.SD15_UNET, torch.tensor(g["sd15_x"]), torch.tensor(g["sd15_t"]),
                         torch.tensor(g["sd15_ctx"]).float(), taps=taps)
    ref = g["sd15_eps"]
    err = np.abs(eps.numpy() - ref).max()
    assert err < 5e-5 * max(1.0, np.abs(ref).max()), err
    for k, v in taps.items():
        flat = v.reshape(-1)
        sample = flat[:: max(1, flat.numel() // 64)][:64].numpy()
        r = g[f"sd15_tap_{k}_sample"]
        assert np.abs(sample - r).max() < 5e-5 * max(1.0, np.abs(r).max()), k


@pytest.mark.skipif(not (GOLD / "golden_sd15.npz").exists(), reason="full-size fixture not generated")
def test_sd15_vae_matches_reference():
    g = dict(np.load(GOLD / "golden_sd15.npz"))
    sd = O.synth_state_dict(O.vae_param_shapes(O.SD15_VAE), seed=22)
    img = O.vae_decode(sd, O.SD15_VAE, torch.tensor(g["sd15_vae_z"]))
    crop = img[:, :, 192:320, 192:320].numpy()
    scale = max(1.0, float(g["sd15_vae_img_stats"][2]))
    assert np.abs(crop - g["sd15_vae_img_crop"]).max() < 5e-5 * scale
    assert np.abs(img[:, :, ::8, ::8].numpy() - g["sd15_vae_img_sub8"]).max() < 5e-5 * scale


def test_tiny_vae_encoder_matches_reference(tiny):
    """Encoder + quant_conv moments and the posterior sample vs the reference's Encoder / DiagonalGaussianDistribution."""
    cfg = O.TINY_VAE
    sd = O.synth_state_dict(O.vae_encoder_param_shapes(cfg), seed=13)
    mom = O.vae_encode_moments(sd, cfg, torch.tensor(tiny["vae_enc_x"]))
    ref = tiny["vae_enc_moments"]
    assert mom.shape == ref.shape
    assert np.abs(mom.numpy() - ref).max() < 2e-5 * np.abs(ref).max()
    z = O.posterior_sample(torch.tensor(ref), torch.tensor(tiny["vae_enc_noise"]), cfg.scale_factor)
    assert np.abs(z.numpy() - tiny["vae_enc_z"]).max() < 1e-6 * np.abs(tiny["vae_enc_z"]).max()


def test_tiny_unet_conv_attention_matches_reference(tiny):
    """Subject-token conv attention (replace_rows_by_conv_attn, 3x3) inside the reference UNet vs the oracle."""
    cfg = O.TINY_UNET
    sd = O.synth_state_dict(O.unet_param_shapes(cfg), seed=11)
    ph = (torch.tensor(tiny["tiny_convattn_idx_b"]), torch.tensor(tiny["tiny_convattn_idx_n"]))
    eps = O.unet_forward(sd, cfg, torch.tensor(tiny["tiny_x"]), torch.tensor(tiny["tiny_t"]), torch.tensor(tiny["tiny_ctx"]),
                         placeholder_indices=ph, conv_attn_kernel_size=3)
    ref = tiny["tiny_convattn_eps"]
    assert np.abs(eps.numpy() - ref).max() < 2e-5 * np.abs(ref).max()
    assert np.abs(ref[0] - tiny["tiny_eps"][0]).max() > 1e-4    # the replacement changes sample 0 ...
    assert np.array_equal(ref[1], tiny["tiny_eps"][1])           # ... and leaves the subject-free sample alone


@pytest.mark.parametrize("case", ["k2", "k4", "multi"])
def test_tiny_unet_conv_attention_kernel_sizes_and_several_strings(tiny, case):
    """Kernel sizes 2 and 4 (asymmetric pads, util.py:747-760) and two subject strings in one batch (attention.py:208-216
    loops over placeholder2indices) inside the reference UNet vs the oracle."""
    cfg = O.TINY_UNET
    sd = O.synth_state_dict(O.unet_param_shapes(cfg), seed=11)
    z = (torch.tensor(tiny["tiny_convattn_idx_b"]), torch.tensor(tiny["tiny_convattn_idx_n"]))
    if case == "k2":
        ph, ks, ref = z, 2, tiny["tiny_convattn_k2_eps"]
    elif case == "k4":
        ph, ks, ref = (torch.tensor(tiny["tiny_convattn_k4_idx_b"]), torch.tensor(tiny["tiny_convattn_k4_idx_n"])), 4, tiny["tiny_convattn_k4_eps"]
    else:
        ph = {"z": z, "y": (torch.tensor(tiny["tiny_convattn_multi_y_idx_b"]), torch.tensor(tiny["tiny_convattn_multi_y_idx_n"]))}
        ks, ref = 3, tiny["tiny_convattn_multi_eps"]
    eps = O.unet_forward(sd, cfg, torch.tensor(tiny["tiny_x"]), torch.tensor(tiny["tiny_t"]), torch.tensor(tiny["tiny_ctx"]),
                         placeholder_indices=ph, conv_attn_kernel_size=ks)
    assert np.abs(eps.numpy() - ref).max() < 2e-5 * np.abs(ref).max()
    assert np.abs(ref - tiny["tiny_convattn_eps"]).max() > 1e-4      # each case differs from the single-string 3x3 one


def test_tiny_ddim_inpainting_matches_reference_sampler(tiny):
    """The mask / x0 blend of ddim_sampling (ddim.py:190-195) with the recorded q_sample noise."""
    cfg = O.TINY_UNET
    sd = O.synth_state_dict(O.unet_param_shapes(cfg), seed=11)
    apply = lambda x, t, c: O.unet_forward(sd, cfg, x, t, c)
    out = O.ddim_sample(apply, O.register_schedule(), 5, torch.tensor(tiny["ddim_xT"]), torch.tensor(tiny["ddim_c"]),
                        torch.tensor(tiny["ddim_uc"]), guidance_scale=(8.0, 3.0), mask=torch.tensor(tiny["inpaint_mask"]),
                        x0=torch.tensor(tiny["inpaint_x0"]), q_noise=torch.tensor(tiny["inpaint_q_noise"]))
    ref = tiny["inpaint_S5_samples"]
    assert np.abs(out.numpy() - ref).max() < 1e-5 * np.abs(ref).max()


def test_tiny_ddim_score_corrector_and_quantize_match_reference_sampler(tiny):
    """p_sample_ddim's score_corrector (ddim.py:262-264) and quantize_denoised (:281-282) branches: the reference sampler was
    run with the generator's corrector (0.9 e_t + 0.05 x) and quantiser (round(4 z) / 4), alone and together."""
    cfg = O.TINY_UNET
    sd = O.synth_state_dict(O.unet_param_shapes(cfg), seed=11)
    apply = lambda x, t, c: O.unet_forward(sd, cfg, x, t, c)
    corr = lambda e, x, t: 0.9 * e + 0.05 * x
    quant = lambda z: torch.round(z * 4.0) / 4.0
    for key, kw in (("corr_S5_samples", dict(score_corrector=corr)), ("quant_S5_samples", dict(quantize=quant)),
                    ("corrquant_S5_samples", dict(score_corrector=corr, quantize=quant))):
        out = O.ddim_sample(apply, O.register_schedule(), 5, torch.tensor(tiny["ddim_xT"]), torch.tensor(tiny["ddim_c"]),
                            torch.tensor(tiny["ddim_uc"]), guidance_scale=(7.0, 3.0), **kw)
        ref = tiny[key]
        # (a quantiser is discontinuous: an fp32 rounding difference that crosses a step moves that element by 1/4.  None does here.)
        assert np.abs(out.numpy() - ref).max() < 1e-5 * np.abs(ref).max(), key


def test_tiny_unet_compel_cfg_matches_reference(tiny):
    cfg = O.TINY_UNET
    sd = O.synth_state_dict(O.unet_param_shapes(cfg), seed=11)
    eps = O.unet_forward(sd, cfg, torch.tensor(tiny["tiny_x"]), torch.tensor(tiny["tiny_t"]), torch.tensor(tiny["tiny_ctx"]),
                         compel_cfg=(torch.tensor(tiny["tiny_compel_empty"]), 2.0))
    ref = tiny["tiny_compel_eps"]
    assert np.abs(eps.numpy() - ref).max() < 2e-5 * np.abs(ref).max()
    assert np.array_equal(ref[1], tiny["tiny_eps"][1])   # only the first half of the batch is re-weighted


def test_sd15_vae_encoder_matches_reference():
    g = np.load(GOLD / "golden_sd15.npz")
    cfg = O.SD15_VAE
    sd = O.synth_state_dict(O.vae_encoder_param_shapes(cfg), seed=23)
    assert sum(int(np.prod(v.shape)) for v in sd.values()) == 34_163_592 + 72   # Encoder + quant_conv
    x = torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(int(g["sd15_enc_x_seed"][0]))) * 2.0 - 1.0
    mom = O.vae_encode_moments(sd, cfg, x)
    ref = g["sd15_enc_moments"]
    assert np.abs(mom.numpy() - ref).max() < 5e-5 * np.abs(ref).max()


def test_reference_module_goldens(tiny):
    """outputs of the reference's own GroupNorm32 / Normalize / LayerNorm / FeedForward(GEGLU) /
    CrossAttention (self N=64 dh=160, cross S=77) / ResBlock / Downsample / Upsample / SpatialTransformer (context
    passed as the layerwise callable) vs the oracle's functions on the same seeded weights."""
    import torch.nn.functional as F
    sys.path.insert(0, str(GOLD))
    from opgold import module_params as mp
    T = lambda k: torch.tensor(tiny[k])

    def close(got, key, tol=2e-5):
        ref = tiny[key]
        assert got.shape == ref.shape, (key, got.shape, ref.shape)
        assert np.abs(got.numpy() - ref).max() < tol * np.abs(ref).max(), key

    close(F.silu(O._gn(mp("gn32", "n."), "n", T("op_gn_x"), 1e-5)), "op_gn32_silu")
    close(O._gn(mp("normalize", "n."), "n", T("op_gn_x"), 1e-6), "op_normalize")
    close(O._ln(mp("ln", "n."), "n", T("op_ln_x")), "op_ln")
    close(O.feed_forward(mp("ff", "f."), "f", T("op_ln_x")), "op_ff_geglu")
    close(O.cross_attention(mp("attn_self", "a."), "a", T("op_attn_self_x"), None, None, 8), "op_attn_self")
    ctx = T("op_attn_cross_ctx")
    close(O.cross_attention(mp("attn_cross", "a."), "a", T("op_attn_cross_x"), ctx, ctx, 4), "op_attn_cross")
    x, emb = T("op_res_x"), T("op_res_emb")
    close(O.resblock(mp("res_same", "r."), "r", x, emb), "op_resblock_same")
    close(O.resblock(mp("res_widen", "r."), "r", x, emb), "op_resblock_widen")
    close(O._conv(mp("down", "d."), "d.op", x, stride=2), "op_downsample")
    close(O._conv(mp("up", "u."), "u.conv", F.interpolate(x, scale_factor=2, mode="nearest")), "op_upsample")
    close(O.spatial_transformer(mp("st", "s."), "s", x, T("op_st_ctx"), 2, 1), "op_spatial_transformer")


# ------------------------------------------------------------------ CLIP text tower ----------
def test_clip_oracle_matches_transformers_golden():
    """oracle/clip_oracle.py against a randomly initialised transformers.CLIPTextModel driven like the reference's
    patched forwards (tests/golden/gen_golden_clip.py): plain prompts, another last-layers blend, a batch whose token
    embeddings the EmbeddingManager restatement patched (48 = 3 x 16 layer copies), and the full-size tower."""
    from oracle import clip_oracle as CO
    g = dict(np.load(GOLD / "golden_clip.npz"))
    cfg = CO.TINY_CLIP
    sd = O.synth_state_dict(CO.clip_param_shapes(cfg), seed=41)
    ids = torch.tensor(g["tiny_ids"])
    z = CO.clip_text_forward(sd, cfg, CO.clip_embed_tokens(sd, ids))
    assert np.abs(z.numpy() - g["tiny_z"]).max() < 2e-5 * np.abs(g["tiny_z"]).max()
    z = CO.clip_text_forward(sd, cfg, CO.clip_embed_tokens(sd, ids), skip_weights=(0.2, 0.8))
    assert np.abs(z.numpy() - g["tiny_z_w28"]).max() < 2e-5 * np.abs(g["tiny_z_w28"]).max()
    assert np.abs(g["tiny_z_w28"] - g["tiny_z"]).max() > 1e-3          # the blend weights matter
    ids_p = torch.tensor(g["tiny_ids_subj"])
    patched, ph, mask = CO.embedding_manager_patch(ids_p, CO.clip_embed_tokens(sd, ids_p), 777, torch.tensor(g["tiny_subj_emb"]))
    z = CO.clip_text_forward(sd, cfg, patched)
    assert z.shape == (48, 77, 64) and np.abs(z.numpy() - g["tiny_z_subj"]).max() < 2e-5 * np.abs(g["tiny_z_subj"]).max()
    cfg = CO.SD15_CLIP
    sd = O.synth_state_dict(CO.clip_param_shapes(cfg), seed=42)
    z = CO.clip_text_forward(sd, cfg, CO.clip_embed_tokens(sd, torch.tensor(g["sd15_ids"])))
    assert np.abs(z.numpy() - g["sd15_z"]).max() < 5e-5 * np.abs(g["sd15_z"]).max()
    assert sum(int(np.prod(s)) for s in CO.clip_param_shapes(cfg).values()) == 123_060_480   # CLIP ViT-L/14 text tower


def test_clip_oracle_zero_shot_identity_path_matches_transformers_golden():
    """the two CLIP-tower drives of the zero-shot identity path (plain last state; last three states
    weighted [1, 2, 4] / 7) against the randomly initialised transformers.CLIPTextModel of tests/golden/gen_golden_clip.py."""
    import numpy as np
    import torch
    from oracle import clip_oracle as CO
    from oracle import ldm_oracle as O
    g = dict(np.load(GOLD / "golden_clip.npz"))
    cfg = CO.TINY_CLIP
    sd_t = O.synth_state_dict(CO.clip_param_shapes(cfg), seed=41)
    emb = CO.clip_embed_tokens(sd_t, torch.tensor(g["tiny_ids"]))
    for key, w in (("tiny_z_w1", (1.0,)), ("tiny_z_w124", (1.0, 2.0, 4.0))):
        z = CO.clip_text_forward(sd_t, cfg, emb, skip_weights=w)
        assert np.abs(z.numpy() - g[key]).max() < 2e-5 * max(1.0, np.abs(g[key]).max()), key
    sd_a = O.synth_state_dict(CO.clip_param_shapes(cfg), seed=43)
    sd_p = O.synth_state_dict(CO.clip_param_shapes(cfg), seed=44)
    full_a, core = CO.arc2face_forward_face_embs(sd_a, cfg, torch.tensor(g["zs_ids_arc2face"]), 333, torch.tensor(g["zs_face"]))
    assert np.abs(full_a.numpy() - g["zs_arc2face_full"]).max() < 2e-5 * np.abs(g["zs_arc2face_full"]).max()
    pad = CO.clip_pad_embeddings(sd_p, cfg, 1)
    full_p, half, core_p = CO.arc2face_inverse_face_prompt_embs(sd_p, cfg, torch.tensor(g["zs_ids_inverse"]), core, pad)
    assert np.abs(full_p.numpy() - g["zs_inverse_full"]).max() < 2e-5 * np.abs(g["zs_inverse_full"]).max()
    # properties of the un-pinned generator restatement: shape, layer copies identical at scale 1, pads at scale 0
    zs, half2 = CO.subj_basis_generator_face(sd_p, cfg, torch.tensor(g["zs_ids_inverse"]), core, 1, out_id_embs_scale=1.0)
    assert zs.shape == (2, 16, 16, cfg.hidden) and torch.equal(zs[:, 0], zs[:, 7]) and torch.allclose(zs[:, 3], core_p)
    zs0, _ = CO.subj_basis_generator_face(sd_p, cfg, torch.tensor(g["zs_ids_inverse"]), core, 1, out_id_embs_scale=0.0)
    assert torch.allclose(zs0[1, 5], pad[2:18]) and torch.equal(half2[:, 24:50], pad[24:50].expand(2, -1, -1))
