"""CPU-only tests: host logic of the drop-in classes, the C-ABI surface (load + symbols, no compute),
and the N>1 sharding / gather path over gloo with world_size 2."""
import ctypes
import os
import re
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import ldm_oracle as O  # noqa: E402

GOLD = ROOT / "tests" / "golden"


# ------------------------------------------------------------------ C ABI -----------------------
def _header_symbols():
    text = (ROOT / "include" / "adaface_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(af_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from adaface_amd import _lib
    assert _header_symbols() == sorted(_lib.EXPORTED_SYMBOLS)


def test_library_loads_and_exports_every_symbol():
    """The in-tree .so must load (HIP runtime present, no GPU needed) and export everything the header declares."""
    from adaface_amd import _lib
    if not _lib.lib_path().exists():
        from adaface_amd import build
        build.build(verbose=False)
    lib = ctypes.CDLL(os.fspath(_lib.lib_path()))
    for sym in _header_symbols():
        assert hasattr(lib, sym), sym
    assert _lib.load().af_version() >= 1


def test_no_wide_store_is_followed_by_a_write_of_its_data_registers():
    """gfx950 code of every translation unit: no 12/16-byte VMEM store has a data register rewritten by a VALU instruction in
    the next two slots (hipcc leaves the SGPR-offset form unpadded; seen to store the new value, scripts/check_isa_hazards.py)."""
    import importlib.util
    from adaface_amd import _lib, build
    if not _lib.lib_path().exists() or not list((ROOT / "adaface_amd" / "_build").glob("*.o")):
        build.build(verbose=False)        # (a fresh tree only; objects OLDER than their sources make scan() raise, not rebuild)
    spec = importlib.util.spec_from_file_location("check_isa_hazards", ROOT / "scripts" / "check_isa_hazards.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    stores, hits = mod.scan(slots=2)
    assert stores > 500, stores          # (the scan saw the kernels: ~1800 wide stores in the library)
    assert not hits, hits


def test_no_cpu_fallback_in_product():
    """The product package must not import the oracle, and must refuse CPU tensors."""
    for path in (ROOT / "adaface_amd").rglob("*.py"):
        src = path.read_text()
        assert "import oracle" not in src and "from oracle" not in src, path
    from adaface_amd import ops
    with pytest.raises(ValueError):
        ops.group_norm(torch.zeros(1, 32, 4, 4), torch.ones(32), torch.zeros(32))


# ------------------------------------------------------------------ drop-in classes --------------
def test_dropin_state_dict_keys_match_reference_inventory():
    from adaface_amd.configs import sd15_config, tiny_config
    from ldm.util import instantiate_from_config
    m = instantiate_from_config(tiny_config()["model"])
    keys = {k: tuple(v.shape) for k, v in m.state_dict().items() if k.startswith(("model.", "first_stage_model."))}
    ref = dict(O.unet_param_shapes(O.TINY_UNET))
    ref.update(O.vae_param_shapes(O.TINY_VAE))
    ref.update(O.vae_encoder_param_shapes(O.TINY_VAE))   # encoder.* + quant_conv (init-image side)
    assert keys == {k: tuple(v) for k, v in ref.items()}
    # SD-1.5 inventory without allocating 3.4 GB: through the layout module
    from adaface_amd import layout
    p = sd15_config()["model"]["params"]["unet_config"]["params"]
    shapes = layout.unet_param_shapes(**p)
    assert len(shapes) == 686 and sum(int(np.prod(s)) for s in shapes.values()) == 859_520_964
    assert len(layout.unet_zero_init_names(shapes)) == 2 * 39  # weight + bias of the 39 zero_module layers
    # load_state_dict with the reference key names works and marks the engine weights dirty
    sd = O.synth_state_dict(ref, seed=3)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(not k.startswith(("model.", "first_stage_model.")) for k in missing)
    assert m.model.diffusion_model._weights_dirty and m.first_stage_model._weights_dirty


def test_forward_on_cpu_raises_loudly():
    from adaface_amd.configs import tiny_config
    from ldm.util import instantiate_from_config
    m = instantiate_from_config(tiny_config()["model"])
    x = torch.zeros(1, 4, 16, 16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.apply_model(x, torch.tensor([1]), m.get_learned_conditioning(torch.zeros(16, 77, 64)))
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.decode_first_stage(x)


def test_sampler_schedule_matches_reference_golden():
    from adaface_amd.configs import tiny_config
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.util import instantiate_from_config
    g = dict(np.load(GOLD / "golden_tiny.npz"))
    m = instantiate_from_config(tiny_config()["model"])
    np.testing.assert_array_equal(m.betas.numpy(), g["sched_betas"].astype(np.float32))
    np.testing.assert_array_equal(m.alphas_cumprod.numpy(), g["sched_alphas_cumprod"].astype(np.float32))
    s = DDIMSampler(m)
    for S in (10, 50):
        s.make_schedule(S, verbose=False)
        np.testing.assert_array_equal(s.ddim_timesteps, g[f"ddim_timesteps_S{S}"])
        np.testing.assert_array_equal(s.ddim_alphas.numpy(), g[f"ddim_alphas_S{S}"])
        np.testing.assert_array_equal(s.ddim_alphas_prev, g[f"ddim_alphas_prev_S{S}"])
        np.testing.assert_array_equal(s.ddim_sigmas, g[f"ddim_sigmas_S{S}"])


def test_unet_rejects_out_of_scope_options():
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    kw = dict(image_size=32, in_channels=4, model_channels=64, out_channels=4, num_res_blocks=2,
              attention_resolutions=[4, 2, 1], channel_mult=[1, 2, 4, 4], num_heads=2, use_spatial_transformer=True,
              context_dim=64)
    UNetModel(**kw)
    with pytest.raises(NotImplementedError):
        UNetModel(**{**kw, "num_classes": 10})
    with pytest.raises(NotImplementedError):
        UNetModel(**{**kw, "use_scale_shift_norm": True})


# ------------------------------------------------------------------ N > 1 (gloo, CPU) -------------
def test_shard_range_covers_batch_exactly_once():
    from adaface_amd.parallel import shard_range
    for gb in (1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            seen = []
            for r in range(w):
                lo, hi = shard_range(gb, r, w)
                seen += list(range(lo, hi))
            assert seen == list(range(gb)), (gb, w)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _gloo_worker(rank, world, port, ragged, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from adaface_amd.parallel import gather_frames, shard_batch
        gb = 5 if ragged else 4
        g = torch.Generator().manual_seed(42)  # every rank draws the same GLOBAL tensors
        x_T = torch.randn(gb, 4, 8, 8, generator=g)
        ctx = torch.randn(gb * 16, 7, 8, generator=g)
        mine = shard_batch(x_T, rank, world)
        cmine = shard_batch(ctx, rank, world, per_sample=16)
        assert cmine.shape[0] == mine.shape[0] * 16
        # stand-in for sample+decode: a deterministic per-sample function of the inputs
        frames = (mine.sum(dim=(1, 2, 3), keepdim=False)[:, None, None, None] * torch.ones(1, 6, 6, 3)).mul(10).to(torch.uint8)
        out = gather_frames(frames, global_batch=gb)
        ref = (x_T.sum(dim=(1, 2, 3))[:, None, None, None] * torch.ones(1, 6, 6, 3)).mul(10).to(torch.uint8)
        q.put((rank, bool(torch.equal(out, ref)), tuple(out.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ragged", [False, True])
def test_gloo_world2_shard_and_gather(ragged):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, ragged, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(shape[0] == (5 if ragged else 4) for _, _, shape in res)


# ------------------------------------------------------------------ bench.py's own launcher path (gloo, CPU) ----
def _run_bench(args, env=None, timeout=300):
    import json
    import subprocess
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None), e.pop("RANK", None), e.pop("LOCAL_RANK", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, env=e, timeout=timeout)
    line = next((l for l in reversed(r.stdout.splitlines()) if l.startswith("{")), None)
    return r, (json.loads(line) if line else None)


def test_bench_launches_its_own_ranks_gloo_world2():
    """`python bench.py --gpus 2` with no launcher in the environment must start two ranks itself (fresh processes under
    torch.distributed.run), report the number of ranks the group connected, and produce the same gathered frames as one
    rank — driven here with the HIP path stubbed out (--plumbing-test, gloo): launcher, rendezvous on 127.0.0.1,
    shard_batch, micro-batching, gather_frames, max-over-ranks timing and the JSON line are bench.py's real code."""
    common = ["--plumbing-test", "--steps", "2", "--warmup", "1", "--batch", "2", "--global-batch", "8"]
    r2, j2 = _run_bench(["--gpus", "2", *common])
    assert r2.returncode == 0, r2.stderr[-2000:]
    r1, j1 = _run_bench(["--gpus", "1", *common])
    assert r1.returncode == 0, r1.stderr[-2000:]
    assert j2["n_gpus"] == 2 and j1["n_gpus"] == 1
    assert j2["config"]["global_batch"] == 8 and j2["scaling"] == "strong" and j2["value"] is None
    assert j2["checksum"] == j1["checksum"]            # results do not depend on the world size
    assert j2["frames_sha256"] == j1["frames_sha256"]  # ... byte for byte: the rank-sharded, gathered frames ARE the world-1 frames
    # per-rank view for reading a scaling curve: every rank's own wall time (min / max) and the all-gather's duration field
    pr = j2["distributed"]["per_rank"]
    assert set(pr) == {"dt_s_min", "dt_s_max", "all_gather_ms_per_step_mean", "all_gather_ms_per_step_max_over_ranks"}
    assert 0 < pr["dt_s_min"] <= pr["dt_s_max"] and abs(pr["dt_s_max"] * 1e3 / 2 - j2["ms_per_step"]) < 1e-6
    assert j2["distributed"]["world_size"] == 2 and j2["distributed"]["all_gather_calls"] == 3   # warm-up + 2 steps
    assert "per_rank" in j1["distributed"]
    assert "PLUMBING TEST" in j2["metric"]


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """--gpus 4 inside a 2-rank launch must fail instead of silently measuring fewer GPUs."""
    r, j = _run_bench(["--gpus", "4", "--plumbing-test", "--steps", "1", "--warmup", "0"],
                      env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                           "MASTER_PORT": str(_free_port())}, timeout=120)
    assert r.returncode != 0 and j is None
    assert "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_identity_context_places_the_padded_id_vector_in_rows_4_to_19():
    """BASELINE config 4's synthetic context: one unit-norm 512-d identity vector per sample, zero-padded to
    768 (util.py:1111), in rows 4..19 of EVERY layer copy; the other rows are the plain prompt context."""
    from adaface_amd.synth import synth_context, synth_context_identity
    c = synth_context_identity(3, seed=100, device="cpu").reshape(3, 16, 77, 768)
    base = synth_context(3, seed=100, device="cpu").reshape(3, 16, 77, 768)
    assert torch.equal(c[:, :, :4], base[:, :, :4]) and torch.equal(c[:, :, 20:], base[:, :, 20:])
    ident = c[:, 0, 4]
    assert (c[:, :, 4:20] == ident[:, None, None, :]).all()          # the same vector in all 16 rows of all 16 layers
    assert (ident[:, 512:] == 0).all() and torch.allclose(ident[:, :512].norm(dim=1), torch.full((3,), 512 ** 0.5))
    assert not torch.equal(ident[0], ident[1])                        # a different identity per sample


def test_bench_config4_plumbing_is_strong_scaling_batch_64():
    """`bench.py --workload config4` (identity context, fp8 mode, global batch 64 in micro-batches) through the stubbed path."""
    r, j = _run_bench(["--gpus", "1", "--workload", "config4", "--plumbing-test", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert j["config"]["global_batch"] == 64 and j["scaling"] == "strong" and "identity" in j["config"]["workload"]


def test_adaprompt_context_differs_per_layer_only_in_subject_rows():
    from adaface_amd.synth import synth_context, synth_context_adaprompt
    c = synth_context_adaprompt(3, seed=100, device="cpu").reshape(3, 16, 77, 768)
    base = synth_context(3, seed=100, device="cpu").reshape(3, 16, 77, 768)
    assert torch.equal(c[:, :, :6], base[:, :, :6]) and torch.equal(c[:, :, 22:], base[:, :, 22:])
    assert (c[:, 0, :6] == c[:, 5, :6]).all()                      # plain rows identical over the 16 layer copies
    assert not torch.equal(c[:, 0, 6:22], c[:, 1, 6:22])           # subject rows drawn per layer
    assert abs(c[:, :, 6:22].std().item() - 0.07) < 0.005


# ------------------------------------------------------------------ EmbeddingManager, inference subset (parity unpinned) ----
def test_embedding_manager_forward_properties():
    """The product's EmbeddingManager.forward against the properties the reference's source states
    (embedding_manager.py:1292-1586) and against the oracle's independent restatement: 16x tuck with an instance's copies
    adjacent, FIRST occurrence per instance only, K consecutive positions, layer l of the subject vectors into copy l,
    untouched rows elsewhere, placeholder2indices over the original batch, prompt_emb_mask without BOS / EOS-pad."""
    from adaface_amd.ldm.modules.embedding_manager import EmbeddingManager, StaticLayerwiseEmbedding
    from oracle import clip_oracle as CO
    g = torch.Generator().manual_seed(3)
    B, N, D, K, tok = 3, 77, 32, 4, 777
    ids = torch.randint(2, 700, (B, N), generator=g)
    ids[:, 0] = 49406
    ids[:, 60:] = 49407
    ids[0, 5] = tok
    ids[2, 11] = tok
    ids[2, 30] = tok                       # a second occurrence in instance 2: must be left alone
    emb = torch.randn(B, N, D, generator=g)
    subj = torch.randn(16, K, D, generator=g)
    em = EmbeddingManager(None, subject_strings=["z"], num_vectors_per_subj_token=K)
    em.add_placeholder("z", tok, subj)
    out = em(ids, emb)
    assert out.shape == (16 * B, N, D)
    rep = emb.unsqueeze(1).repeat(1, 16, 1, 1).view(16 * B, N, D)
    for l in range(16):
        assert torch.equal(out[0 * 16 + l, 5:5 + K], subj[l])          # instance 0, copy l <- layer l, K consecutive rows
        assert torch.equal(out[2 * 16 + l, 11:11 + K], subj[l])
        assert torch.equal(out[2 * 16 + l, 30], emb[2, 30])             # second occurrence untouched
        assert torch.equal(out[1 * 16 + l], emb[1])                     # instance without the token untouched
    untouched = torch.ones(16 * B, N, dtype=torch.bool)
    untouched[0:16, 5:5 + K] = False
    untouched[32:48, 11:11 + K] = False
    assert torch.equal(out[untouched], rep[untouched])
    idx_b, idx_n = em.placeholder2indices["z"]
    assert idx_b.tolist() == [0] * K + [2] * K and idx_n.tolist() == list(range(5, 5 + K)) + list(range(11, 11 + K))
    assert em.prompt_emb_mask.shape == (B, N, 1) and em.prompt_emb_mask[:, 0].sum() == 0 and em.prompt_emb_mask[:, 60:].sum() == 0
    ref, ph, mask = CO.embedding_manager_patch(ids, emb, tok, subj)
    assert torch.equal(out, ref) and torch.equal(mask, em.prompt_emb_mask)
    assert torch.equal(ph[0], idx_b) and torch.equal(ph[1], idx_n)
    # StaticLayerwiseEmbedding: low-rank combination + per-vector LayerNorm / sqrt(D) + bias  (:500-537)
    r, Npre = 6, 2
    brw, bcw = torch.randn(16, K, r, generator=g) * 0.1, torch.ones(1, K, r) / r
    bv, pv, bias = torch.randn(K, r - Npre, D, generator=g), torch.randn(K, Npre, D, generator=g), torch.randn(16, K, D, generator=g) * 0.01
    sle = StaticLayerwiseEmbedding(brw, bcw, bv, bias, pre_vecs=pv)
    got = sle()
    want = CO.static_layerwise_embedding(brw, bcw, bv, pv, bias)
    assert got.shape == (16, K, D) and torch.allclose(got, want, atol=1e-6)
    assert abs((got - bias).std().item() - 1.0 / D ** 0.5) < 0.02          # LayerNorm'd vectors scaled by 1/sqrt(D)
    em2 = EmbeddingManager(None, subject_strings=["z"])
    em2.add_placeholder("z", tok, sle)
    assert torch.allclose(em2(ids, emb)[0, 5:5 + K], got[0], atol=1e-6)


def test_embedding_manager_tensor_only_checkpoint(tmp_path):
    from adaface_amd.ldm.modules.embedding_manager import EmbeddingManager
    subj = torch.randn(16, 2, 8)
    path = tmp_path / "emb.pt"
    torch.save({"string_to_token": {"z": 123}, "string_to_static_embedder": {"z": subj}, "token2num_vectors": {"z": 2},
                "use_conv_attn_kernel_size": 3, "subject_strings": ["z"], "background_strings": []}, path)
    em = EmbeddingManager(None, subject_strings=[])
    em.load(str(path))
    assert em.string_to_token_dict == {"z": 123} and em.use_conv_attn_kernel_size == 3 and em.token2num_vectors["z"] == 2
    ids = torch.zeros(1, 10, dtype=torch.long)
    ids[0, 4] = 123
    out = em(ids, torch.zeros(1, 10, 8))
    assert torch.equal(out[7, 4:6], subj[7])
    # zero-shot managers build but need the Arc2Face encoder and identity features before they can patch
    zs = EmbeddingManager(None, subject_strings=[], do_zero_shot=True, out_emb_dim=8)
    zs.add_zero_shot_placeholder("z", 123, 2, clip_config=dict(vocab=200, hidden=8, layers=2, heads=2, intermediate=16, max_pos=10))
    with pytest.raises(RuntimeError, match="arc2face_text_encoder"):
        zs(ids, torch.zeros(1, 10, 8))
    with pytest.raises(NotImplementedError):
        from adaface_amd.ldm.modules.subj_basis_generator import SubjBasisGenerator
        SubjBasisGenerator(placeholder_is_bg=True)
