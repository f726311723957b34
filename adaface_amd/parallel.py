"""Data-parallel sampling over the GPUs of one node.

Samples of a batch never interact on the denoising path (GroupNorm statistics are per sample, attention is
per sample and head; the CFG cond/uncond pair of a sample stays on one GPU), so the path shards by sample:
one process per GPU, weights replicated, inputs sliced from globally seeded tensors so results do not depend
on the world size, no communication during the 50 steps, and exactly ONE collective per batch: an all-gather
of the decoded uint8 frames (RCCL over xGMI when the process group backend is "nccl"; "gloo" in CPU tests).
The reference has no equivalent — it is single-process, single-GPU (scripts/stable_txt2img.py:238,342).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


# what the process group did in this process (bench.py reports it: "did RCCL see N ranks" must be checkable from the line)
STATS = {"backend": None, "world_size": 1, "probe_allreduce": None, "all_gather_calls": 0}


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launched_by_torchrun() -> bool:
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def launch_ranks(n_ranks: int, script: str, argv: Sequence[str], env: Optional[dict] = None) -> int:
    """Start `script argv...` as n_ranks FRESH processes (one per GPU) under torch.distributed.run on 127.0.0.1 and
    return its exit code.  Must be called before the calling process has touched the GPU: the ranks are children of
    the elastic launcher, which is a child of ours — no process that initialised HIP is ever replaced by another
    program.  `python bench.py --gpus N` and `scripts/stable_txt2img.py --gpus N` use this when they were not
    themselves started by a launcher (no WORLD_SIZE / RANK in the environment)."""
    if n_ranks < 2:
        raise ValueError("launch_ranks is for n_ranks >= 2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script, *argv]
    e = dict(os.environ if env is None else env)
    # The build / GPU images export HSA_ENABLE_IPC_MODE_LEGACY=0 and document why: the host driver of this pool only supports
    # dmabuf IPC, and without it buffer-handle exchange between processes (RCCL's intra-node transport, CUDA-tensor sharing)
    # fails with `hipIpcGetMemHandle: invalid argument`.  Inherited when set; defaulted for a caller that scrubbed its env.
    # (A world-1 run -- all this build can execute, tests/test_chained_gpu.py -- exchanges no handles and cannot show it.)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=e).returncode


def init_distributed(expected_world: int, backend: str = "nccl", device: Optional[torch.device] = None) -> Tuple[int, int]:
    """Join the process group torch.distributed.run prepared (env://) and CHECK it: returns (rank, world) where world is
    the number of ranks the backend actually connected; raises SystemExit if that differs from `expected_world` or,
    for "nccl" (= RCCL on ROCm), if two ranks share a device.  world == 1 with no launcher: no process group."""
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != expected_world:
        raise SystemExit(f"--gpus {expected_world} but the launcher started WORLD_SIZE={world_env} ranks")
    if world_env == 1 and not launched_by_torchrun():
        return 0, 1
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    # one all-reduce proves every rank is reachable through the backend (and, on GPUs, over xGMI)
    dev = device if backend == "nccl" else torch.device("cpu")
    probe = torch.zeros(world, dtype=torch.int64, device=dev)
    probe[rank] = 1 + (device.index if (backend == "nccl" and device is not None and device.index is not None) else rank)
    dist.all_reduce(probe)
    seen = probe.cpu().tolist()
    STATS.update(backend=backend, world_size=world, probe_allreduce=seen)
    if world != expected_world or any(v == 0 for v in seen):
        raise SystemExit(f"process group has {world} ranks ({seen}), --gpus asked for {expected_world}")
    if backend == "nccl" and len(set(seen)) != world:
        raise SystemExit(f"ranks share a device: local device ids {seen}")
    return rank, world


def world() -> Tuple[int, int]:
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(global_batch: int, rank: Optional[int] = None, world_size: Optional[int] = None) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of the global batch owned by `rank`; the first (global_batch % world) ranks take
    one extra sample, so any global batch is covered exactly once."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, extra = divmod(global_batch, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(t: torch.Tensor, rank: Optional[int] = None, world_size: Optional[int] = None,
                per_sample: int = 1) -> torch.Tensor:
    """Slice dim 0 of a globally generated tensor; `per_sample` = rows per sample (16 for the layerwise context
    [B*16, 77, 768], whose layer copies are adjacent per sample: embedding_manager.py:1342-1353)."""
    if t.shape[0] % per_sample:
        raise ValueError("dim 0 is not a multiple of per_sample")
    lo, hi = shard_range(t.shape[0] // per_sample, rank, world_size)
    return t[lo * per_sample:hi * per_sample]


def gather_frames(frames: torch.Tensor, global_batch: Optional[int] = None) -> torch.Tensor:
    """All-gather decoded frames [b_local, H, W, 3] (uint8) -> [global_batch, H, W, 3] on every rank, in global
    sample order.  Equal shards use one all_gather_into_tensor; ragged shards pad to the largest shard."""
    rank, w = world()
    if w == 1 and not (dist.is_available() and dist.is_initialized()):
        return frames                      # no process group: plain single-process run
    # (a world-1 group still runs the collective: the one-rank launch exercises the same RCCL call as N ranks)
    b_local = frames.shape[0]
    if global_batch is None:
        n = torch.tensor([b_local], device=frames.device, dtype=torch.int64)
        dist.all_reduce(n)
        global_batch = int(n.item())
    sizes = [shard_range(global_batch, r, w) for r in range(w)]
    counts = [hi - lo for lo, hi in sizes]
    if len(set(counts)) == 1:
        out = torch.empty((global_batch,) + tuple(frames.shape[1:]), dtype=frames.dtype, device=frames.device)
        dist.all_gather_into_tensor(out, frames.contiguous())
        STATS["all_gather_calls"] += 1
        return out
    bmax = max(counts)
    padded = torch.zeros((bmax,) + tuple(frames.shape[1:]), dtype=frames.dtype, device=frames.device)
    padded[:b_local] = frames
    bufs = [torch.empty_like(padded) for _ in range(w)]
    dist.all_gather(bufs, padded)
    STATS["all_gather_calls"] += 1
    return torch.cat([bufs[r][:counts[r]] for r in range(w)], dim=0)
