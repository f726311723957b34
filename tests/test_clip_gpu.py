"""GPU parity of the conditioning producer: the CLIP text tower on the HIP kernels, through the C ABI
(af_clip_embed_tokens / af_clip_text_forward) and through the FrozenCLIPEmbedder / EmbeddingManager / LatentDiffusion
drop-ins, against
  (1) tests/golden/golden_clip.npz — a randomly initialised transformers.CLIPTextModel driven like the reference's
      patched forwards (tests/golden/gen_golden_clip.py), and
  (2) the CPU oracle (oracle/clip_oracle.py), itself pinned to (1).
Tolerances relative to max|reference|: f32 mode 2e-4, bf16 mode 3e-2 (twelve pre-LN layers of bf16 GEMMs).
The EmbeddingManager arithmetic itself is PARITY UNPINNED (see its docstring); what is pinned here is the tower given the
patched embeddings."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import clip_oracle as CO  # noqa: E402
from oracle import ldm_oracle as O    # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = ROOT / "tests" / "golden"
TOL = {"f32": 2e-4, "bf16": 3e-2}


def _rel(got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float("inf") if not np.isfinite(got).all() else float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12))


def _clip_kwargs(cfg):
    return dict(vocab=cfg.vocab, hidden=cfg.hidden, layers=cfg.layers, heads=cfg.heads, intermediate=cfg.intermediate,
                max_pos=cfg.max_pos)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("size", ["tiny", "sd15"])
def test_clip_text_tower_golden(gpu, report, dtype, size):
    from adaface_amd.engine import Engine
    g = dict(np.load(GOLD / "golden_clip.npz"))
    cfg, seed = (CO.TINY_CLIP, 41) if size == "tiny" else (CO.SD15_CLIP, 42)
    sd = O.synth_state_dict(CO.clip_param_shapes(cfg), seed=seed)
    eng = Engine(dtype=dtype, clip=_clip_kwargs(cfg))
    assert set(eng.tensor_table()) == set(CO.clip_param_shapes(cfg))
    assert eng.load_state_dict(sd) == []
    ids = torch.tensor(g[f"{size}_ids"], device=gpu)
    emb = eng.clip_embed_tokens(ids)
    ref_emb = CO.clip_embed_tokens(sd, ids.cpu())
    assert _rel(emb.cpu().numpy(), ref_emb.numpy()) < (1e-7 if dtype == "f32" else 5e-3)
    z = eng.clip_text_forward(ref_emb.to(gpu) if dtype == "f32" else emb).cpu().numpy()
    err = _rel(z, g[f"{size}_z"])
    report(f"clip text tower {size} vs transformers golden [{dtype}]", err, float(np.abs(g[f'{size}_z']).max()), TOL[dtype])
    assert err < TOL[dtype], err
    if size == "tiny":
        z28 = eng.clip_text_forward(ref_emb.to(gpu), 0.2, 0.8).cpu().numpy()
        assert _rel(z28, g["tiny_z_w28"]) < TOL[dtype]
    eng.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_dropin_clip_with_embedding_manager(gpu, report, dtype):
    """FrozenCLIPEmbedder.forward(token ids, embedding_manager=...) with the subject vectors injected: [3, 77] ids ->
    [48, 77, 64], against the golden the transformers model produced from the same patched embeddings."""
    from adaface_amd.configs import tiny_config
    from ldm.modules.embedding_manager import EmbeddingManager
    from ldm.modules.encoders.modules import FrozenCLIPEmbedder
    g = dict(np.load(GOLD / "golden_clip.npz"))
    cfg = CO.TINY_CLIP
    sd = O.synth_state_dict(CO.clip_param_shapes(cfg), seed=41)
    clip = FrozenCLIPEmbedder(device=gpu, **tiny_config()["model"]["params"]["cond_stage_config"]["params"])
    missing, unexpected = clip.load_state_dict({k[len("cond_stage_model."):]: v for k, v in sd.items()}, strict=True)
    clip = clip.to(gpu).set_compute_dtype(dtype)
    em = EmbeddingManager(clip, subject_strings=["z"])
    em.add_placeholder("z", 777, torch.tensor(g["tiny_subj_emb"]))
    z = clip(torch.tensor(g["tiny_ids_subj"]), embedding_manager=em)
    assert tuple(z.shape) == (48, 77, 64)
    err = _rel(z.cpu().numpy(), g["tiny_z_subj"])
    report(f"dropin FrozenCLIPEmbedder + EmbeddingManager vs transformers golden [{dtype}]", err, 1.0, TOL[dtype])
    assert err < TOL[dtype], err
    idx_b, idx_n = em.placeholder2indices["z"]
    assert idx_b.tolist() == [0] * 4 + [2] * 4 and idx_n.tolist() == [5, 6, 7, 8, 11, 12, 13, 14]
    # plain prompts through the same module (no manager): [3, 77, 64]
    z0 = clip(torch.tensor(g["tiny_ids"]))
    assert _rel(z0.cpu().numpy(), g["tiny_z"]) < TOL[dtype]


def test_text_to_latent_pipeline_tiny(gpu, report):
    """Token ids -> LatentDiffusion.get_learned_conditioning (CLIP tower + EmbeddingManager, 16x layerwise context and
    extra_info) -> DDIMSampler (5 steps, CFG) in f32 mode, against the oracles chained the same way."""
    from adaface_amd.configs import tiny_config
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.util import instantiate_from_config
    g = dict(np.load(GOLD / "golden_clip.npz"))
    model = instantiate_from_config(tiny_config()["model"]).eval()
    sd = O.synth_state_dict(O.unet_param_shapes(O.TINY_UNET), seed=11)
    sd.update(O.synth_state_dict(CO.clip_param_shapes(CO.TINY_CLIP), seed=41))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and not [k for k in missing if k.startswith(("model.", "cond_stage_model."))]
    model = model.to(gpu).set_compute_dtype("f32")
    subj = torch.tensor(g["tiny_subj_emb"])
    model.embedding_manager.add_placeholder("z", 777, subj)
    ids_c = torch.tensor(g["tiny_ids_subj"])[:2]
    ids_u = torch.tensor(g["tiny_ids"])[:1].repeat(2, 1)
    c = model.get_learned_conditioning(ids_c)
    uc = model.get_learned_conditioning(ids_u)
    assert tuple(c[0].shape) == (32, 77, 64) and c[2]["use_layerwise_context"] and c[2]["placeholder2indices"]["z"] is not None
    x_T = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(8))
    samples, _ = DDIMSampler(model).sample(S=5, conditioning=c, batch_size=2, shape=[4, 16, 16], verbose=False,
                                           guidance_scale=[6.0, 3.0], unconditional_conditioning=uc, eta=0.0, x_T=x_T.to(gpu))
    csd = {k: v for k, v in sd.items() if k.startswith("cond_stage_model.")}
    pc, _, _ = CO.embedding_manager_patch(ids_c, CO.clip_embed_tokens(csd, ids_c), 777, subj)
    pu, _, _ = CO.embedding_manager_patch(ids_u, CO.clip_embed_tokens(csd, ids_u), 777, subj)
    ref_c, ref_u = CO.clip_text_forward(csd, CO.TINY_CLIP, pc), CO.clip_text_forward(csd, CO.TINY_CLIP, pu)
    assert _rel(c[0].cpu().numpy(), ref_c.numpy()) < 2e-4
    ref = O.ddim_sample(lambda x, t, cc: O.unet_forward(sd, O.TINY_UNET, x, t, cc), O.register_schedule(), 5, x_T, ref_c, ref_u,
                        guidance_scale=(6.0, 3.0))
    err = _rel(samples.cpu().numpy(), ref.numpy())
    report("token ids -> CLIP + EmbeddingManager -> 5-step DDIM latent vs chained oracles [f32]", err, float(ref.abs().max()), 1e-3)
    assert err < 1e-3, err


# ---------------------------------------------------------------------------------------------------------------------
# zero-shot identity path
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_clip_tower_three_state_blend_golden(gpu, report, dtype):
    """af_clip_text_forward3: plain last state and the [1, 2, 4] / 7 blend of the last three hidden states
    (CLIPTextModelWrapper.forward with hidden_state_layer_weights, arc2face_models.py:230-243) vs the transformers golden."""
    from adaface_amd.engine import Engine
    g = dict(np.load(GOLD / "golden_clip.npz"))
    cfg = CO.TINY_CLIP
    sd = O.synth_state_dict(CO.clip_param_shapes(cfg), seed=41)
    eng = Engine(dtype=dtype, clip=_clip_kwargs(cfg))
    assert eng.load_state_dict(sd) == []
    emb = CO.clip_embed_tokens(sd, torch.tensor(g["tiny_ids"])).to(gpu)
    for key, w in (("tiny_z_w1", (0.0, 0.0, 1.0)), ("tiny_z_w124", (1 / 7, 2 / 7, 4 / 7)), ("tiny_z", (0.0, 0.5, 0.5))):
        z = eng.clip_text_forward3(emb, *w).cpu().numpy()
        err = _rel(z, g[key])
        report(f"clip tower three-state blend {key} vs transformers golden [{dtype}]", err, float(np.abs(g[key]).max()), TOL[dtype])
        assert err < TOL[dtype], (key, err)
    eng.close()


def _load_wrapper(wrapper, sd):
    missing, unexpected = wrapper.load_state_dict({k[len("cond_stage_model.transformer."):]: v for k, v in sd.items()}, strict=True)
    assert not missing and not unexpected
    return wrapper


def test_zero_shot_identity_path_dropins(gpu, report):
    """arc2face_forward_face_embs / arc2face_inverse_face_prompt_embs through the CLIPTextModelWrapper drop-in against the
    transformers goldens (f32 mode), then SubjBasisGenerator and the EmbeddingManager's zero-shot branch against the CPU
    oracle's restatement (parity unpinned for those two: they are compared with an independent restatement only)."""
    from ldm.modules.arc2face_models import CLIPTextModelWrapper
    from ldm.modules.embedding_manager import EmbeddingManager
    from ldm.util import arc2face_forward_face_embs, arc2face_inverse_face_prompt_embs
    g = dict(np.load(GOLD / "golden_clip.npz"))
    cfg = CO.TINY_CLIP
    kw = _clip_kwargs(cfg)
    sd_a = O.synth_state_dict(CO.clip_param_shapes(cfg), seed=43)
    sd_p = O.synth_state_dict(CO.clip_param_shapes(cfg), seed=44)
    enc = _load_wrapper(CLIPTextModelWrapper(kw).set_compute_dtype("f32"), sd_a).to(gpu)
    face = torch.tensor(g["zs_face"], device=gpu)
    ids_a = torch.tensor(g["zs_ids_arc2face"][:1])
    full, core = arc2face_forward_face_embs(None, enc, face, input_ids=ids_a, arcface_token_id=333)
    err = _rel(full.cpu().numpy(), g["zs_arc2face_full"])
    report("zero-shot: arc2face_forward_face_embs vs transformers golden [f32]", err, 1.0, TOL["f32"])
    assert err < TOL["f32"] and core.shape == (2, 16, cfg.hidden)
    p2t = _load_wrapper(CLIPTextModelWrapper(kw).set_compute_dtype("f32"), sd_p).to(gpu)
    pad = CO.clip_pad_embeddings(sd_p, cfg, 1).to(gpu)
    ids_p = torch.tensor(g["zs_ids_inverse"][:1])
    full_p, core_p = arc2face_inverse_face_prompt_embs(None, p2t, torch.tensor(g["zs_arc2face_full"][:, 4:20], device=gpu), None,
                                                       ["full", "core"], pad, hidden_state_layer_weights=torch.tensor([[1.0], [2.0], [4.0]]),
                                                       input_ids=ids_p)
    err = _rel(full_p.cpu().numpy(), g["zs_inverse_full"])
    report("zero-shot: arc2face_inverse_face_prompt_embs vs transformers golden [f32]", err, 1.0, TOL["f32"])
    assert err < TOL["f32"]

    # ---- the manager's zero-shot branch: prompt ids with the placeholder, ArcFace vector in, patched embeddings out ----
    token, K = 777, 16
    man = EmbeddingManager(do_zero_shot=True, out_emb_dim=cfg.hidden, zs_arc2face_input_ids=ids_a, zs_arcface_token_id=333)
    man.add_zero_shot_placeholder("z", token, K, clip_config=kw, inverse_prompt_input_ids=ids_p, pad_token_id=1)
    gen = man.string_to_subj_basis_generator_dict["z"]
    _load_wrapper(gen.prompt2token_proj.set_compute_dtype("f32"), sd_p)
    man.arc2face_text_encoder = enc
    man = man.to(gpu)
    man.set_zs_image_features(None, face[:1], (0.75, 1.0))
    gen_t = torch.Generator().manual_seed(3)
    ids = torch.randint(2, cfg.vocab, (2, 77), generator=gen_t)
    ids[:, 0] = 0
    ids[0, 5] = token
    ids[1, 9] = token
    emb = torch.randn(2, 77, cfg.hidden, generator=gen_t)
    out = man(ids.to(gpu), emb.to(gpu)).cpu()
    # oracle: the same chain restated on the CPU
    _, core_o = CO.arc2face_forward_face_embs(sd_a, cfg, ids_a, 333, torch.tensor(g["zs_face"][:1]))
    zs, _ = CO.subj_basis_generator_face(sd_p, cfg, ids_p, core_o, 1, out_id_embs_scale=0.75)
    ref, _, _ = CO.embedding_manager_patch(ids, emb, token, zs[0])
    err = _rel(out.numpy(), ref.numpy())
    report("zero-shot: EmbeddingManager(do_zero_shot) patched embeddings vs CPU restatement [f32]", err, 1.0, 1e-3)
    assert out.shape == (32, 77, cfg.hidden) and err < 1e-3, err
