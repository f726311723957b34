"""Chained (multi-step) error of the THROUGHPUT modes at the benchmark's shape, and the RCCL path at world size 1.

VERDICT r2 weak #1: the f32 mode is pinned end to end against the CPU oracle (test_model_gpu.py::test_config0_*), but
nothing failed if the chained bf16 / fp8 drift doubled.  Here BASELINE config 1's shape (SD-1.5, batch 8 -> CFG batch 16,
64x64 latents, annealed guidance [10, 4]) runs S = 10 DDIM steps in bf16 and in fp8 mode against the f32 mode of the SAME
batch (same synthetic weights as bench.py).  The reference has no bf16 / fp8 path, so the bars are this package's stated
tolerances, relative to max|final latent| of the f32 mode:

    per-forward (first step's eps, cond half): bf16 <= 3e-2 (max-abs, as test_model_gpu's BF16_FWD_BAR), fp8 <= 8e-2
                                               (as test_fp8_gpu states)
    final latent after the chain:                                  bf16 <= 3e-2, fp8 <= 1e-1
    and, derived from the measured per-forward error e1 of the same run:  final <= CHAIN_GAIN * e1  (the chain may not
    amplify the per-forward error by more than the stated gain; measured gains are written to parity_report.txt)

Round 4 adds the RMS of the final deviation (relative to the rms of the f32 latent) under the same numbers.  The max-abs
statistic is a draw of the rounding noise: scaling x_T by 1 + k * 1e-6 (k = 0..5) moves it over 2.68e-2 .. 3.05e-2 for ONE
build (scripts/lab/chained_sensitivity.py; a kernel that only changes an fp32 summation order does the same), while the rms
stays within 2.62e-2 .. 2.64e-2 -- a change of the rms is a change of the arithmetic, a change of the max-abs need not be.
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

pytestmark = pytest.mark.gpu

FWD_BAR = {"bf16": 3e-2, "fp8": 8e-2}
FINAL_BAR = {"bf16": 3e-2, "fp8": 1e-1}
FINAL_RMS_BAR = {"bf16": 3e-2, "fp8": 1e-1}     # rms(final - f32 final) / rms(f32 final): the stable statistic (see above)
CHAIN_GAIN = 2.5


@pytest.fixture(scope="module")
def bench_model(gpu):
    from bench import build_model
    return build_model(gpu, "f32")


# BASELINE.json configs 1 / 2 / 4 differ on this path only in the conditioning they feed: the plain prompt context, the
# AdaPrompt subject rows (per-layer-different rows 6..21) and the identity rows 4..19 (config 4 also names the fp8 mode)
@pytest.mark.parametrize("workload", ["config1", "config2", "config4"])
def test_config1_shape_chained_error_bf16_fp8_vs_f32(gpu, report, bench_model, workload):
    from adaface_amd import synth
    from adaface_amd.synth import synth_context
    from ldm.models.diffusion.ddim import DDIMSampler
    B, S = 8, 10
    model = bench_model
    g = torch.Generator().manual_seed(42)
    x_T = torch.randn(B, 4, 64, 64, generator=g).to(gpu)
    make_ctx = {"config1": synth.synth_context, "config2": synth.synth_context_adaprompt, "config4": synth.synth_context_identity}[workload]
    c_emb = make_ctx(B, seed=100, device=gpu)
    uc_emb = synth_context(B, seed=101, device=gpu, shared=True)
    sampler = DDIMSampler(model)
    t0 = torch.full((B,), 901, dtype=torch.long, device=gpu)
    out = {}
    for mode in ("f32", "bf16", "fp8"):
        model.set_compute_dtype(mode)
        c = model.get_learned_conditioning(c_emb)
        uc = model.get_learned_conditioning(uc_emb)
        eps_c = model.apply_model(x_T, t0, c)                       # the first forward of the chain, cond half
        lat, _ = sampler.sample(S=S, conditioning=c, batch_size=B, shape=[4, 64, 64], verbose=False,
                                guidance_scale=[10.0, 4.0], unconditional_conditioning=uc, eta=0.0, x_T=x_T)
        torch.cuda.synchronize()
        assert torch.isfinite(lat).all() and torch.isfinite(eps_c).all(), mode
        out[mode] = (eps_c.clone(), lat.clone())
    e_scale = out["f32"][0].abs().max().item()
    l_scale = out["f32"][1].abs().max().item()
    for mode in ("bf16", "fp8"):
        e1 = (out[mode][0] - out["f32"][0]).abs().max().item() / e_scale
        ef = (out[mode][1] - out["f32"][1]).abs().max().item() / l_scale
        report(f"{workload} shape Bf=16: first-forward eps {mode} vs f32 mode", e1, e_scale, FWD_BAR[mode])
        report(f"{workload} shape Bf=16: final latent after S=10 DDIM steps {mode} vs f32 mode", ef, l_scale, FINAL_BAR[mode])
        report(f"{workload} shape Bf=16: chain gain (final / first-forward) {mode}", ef / e1, 1.0, CHAIN_GAIN)
        d = (out[mode][1] - out["f32"][1]).double()
        l_rms = out["f32"][1].double().pow(2).mean().sqrt().item()
        erms = d.pow(2).mean().sqrt().item() / l_rms
        report(f"{workload} shape Bf=16: final latent after S=10 DDIM steps {mode} vs f32 mode, rms / rms", erms, l_rms, FINAL_RMS_BAR[mode])
        assert erms <= FINAL_RMS_BAR[mode], (mode, erms)
        assert e1 <= FWD_BAR[mode], (mode, e1)
        assert ef <= FINAL_BAR[mode], (mode, ef)
        assert ef <= CHAIN_GAIN * e1, (mode, ef, e1)
    # the modes really differ (a silent fall-back to one path would make the bars vacuous)
    assert (out["bf16"][1] - out["f32"][1]).abs().max().item() > 0
    assert (out["fp8"][1] - out["bf16"][1]).abs().max().item() > 0


def test_bench_under_torchrun_world1_runs_rccl(gpu):
    """`init_process_group("nccl")` (= RCCL), the proving all-reduce and all_gather_into_tensor of
    adaface_amd/parallel.py executed once on a real MI355X, at world size 1: bench.py as a FRESH child under
    torch.distributed.run (the launcher starts before anything in that child touches the GPU; this pytest process only
    spawns it).  Asserts rc 0, n_gpus == 1 and that the line says the collective path ran."""
    from adaface_amd.parallel import free_port
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.fspath(ROOT / "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline", "--no-parity-leg", "--no-kernel-timing", "--no-inflight-leg"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["dtype"] == "bf16", res
    d = res["distributed"]
    assert d["backend"] == "nccl" and d["world_size"] == 1 and d["probe_allreduce"] == [1], d
    assert d["all_gather_calls"] == 1, d
    # the all-gather's own HIP-event duration and the per-rank wall time are on the line (they separate compute skew from
    # the collective once N > 1 exists)
    pr = d["per_rank"]
    assert pr["all_gather_ms_per_step_mean"] is not None and pr["all_gather_ms_per_step_mean"] >= 0 and pr["dt_s_max"] >= pr["dt_s_min"] > 0, pr
