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
