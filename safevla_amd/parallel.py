"""Data-parallel plumbing: one process per GPU, environments sharded over ranks, RCCL over xGMI.

The hot path shards by environment index only (rows of one env are coupled through the GAE scan and the decoder's
time axis; different envs are independent) -- mirroring the reference's split of ``num_train_processes`` over devices
(/root/reference/training/online/base.py:194-224: evenly_distribute_count_into_bins; device of env i = devices[i % n],
:93-97).  Exchange steps per optimiser step:
  1. ONE all-reduce (SUM) of the flat fp32 gradient arena.  Every rank's loss kernels already scale by
     1 / (global rows of the minibatch), so SUM == the reference engine's "weight by local/global batch size,
     then sum" [3P AllenAct] without a separate scaling pass.  (Upstream all-reduces ~417 tensors one by one,
     including 3 x 35 M frozen-T5 zeros.)
  2. once per rollout: all-reduce of [sum episode cost, #episodes] -> identical Jc on every rank -> replicated
     deterministic lambda update (no broadcast).
Device-agnostic on purpose (works on gloo/CPU tensors) so the N>1 path is covered by world_size-2 CPU tests.
"""
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def evenly_distribute_count_into_bins(count: int, nbins: int) -> List[int]:
    """AllenAct helper used at training/online/base.py:208-224: first (count % nbins) bins get one extra."""
    q, r = divmod(count, nbins)
    return [q + 1 if i < r else q for i in range(nbins)]


def shard_envs(num_envs: int, world: int, rank: int) -> Tuple[int, int]:
    bins = evenly_distribute_count_into_bins(num_envs, world)
    return sum(bins[:rank]), bins[rank]


def force_dist() -> bool:
    """SVLA_FORCE_DIST=1: a SINGLE rank still initialises the backend and sends every exchange step through its collectives (RCCL
    communicator, asynchronous handles on the tower streams, fp64 reductions) -- the 1-GPU test boxes' way to execute the RCCL code path
    (tests/test_dp_gpu.py::test_single_rank_through_rccl); results must equal the non-distributed run bit for bit."""
    return os.environ.get("SVLA_FORCE_DIST") == "1"


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_dist())


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force_dist()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # "nccl" IS RCCL on ROCm.  SVLA_DIST_BACKEND=gloo lets several ranks share one GPU (1-GPU test boxes).
            backend = os.environ.get("SVLA_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if torch.cuda.is_available():
        local = local % torch.cuda.device_count()
    return rank, local, world


def broadcast_model_(model, src: int = 0):
    """Identical replicas: broadcast the trainable arena AND every other parameter / buffer (the frozen T5 text encoder is drawn from
    the global RNG at construction, so ranks seeded differently would otherwise hold different frozen encoders), then refresh the
    bf16 mirrors.  No-op on a single rank."""
    if not is_dist():
        return
    ar = model.arena
    dist.broadcast(ar.flat_p, src=src)
    lo, hi = ar.flat_p.data_ptr(), ar.flat_p.data_ptr() + ar.flat_p.numel() * ar.flat_p.element_size()
    seen = set()
    for t in list(model.parameters()) + list(model.buffers()):
        if lo <= t.data_ptr() < hi or t.data_ptr() in seen:       # arena views / tied tensors
            continue
        seen.add(t.data_ptr())
        dist.broadcast(t.data, src=src)
    model.sync_weights()


def allreduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    if is_dist():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


class _Done:
    def wait(self):
        return True


class _Bf16Exchange:
    """The gradient range travels as bf16 (half the bytes on the xGMI rings: 42 MB per tower instead of 84): every rank rounds its
    local sum to bf16, the backend sums the bf16 values, ``wait()`` writes the result back into the fp32 range.  Opt-in (PPOLagConfig.grad_allreduce_dtype):
    the fp32 exchange is bit-identical to the single-process sum of the shards, this one is within bf16 rounding of it (tests/test_parallel_cpu.py)."""

    def __init__(self, t: torch.Tensor, group=None):
        self.t = t
        self.buf = t.to(torch.bfloat16)
        self.h = dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=group, async_op=True)

    def wait(self):
        self.h.wait()
        self.t.copy_(self.buf)
        return True


def allreduce_sum_async(t: torch.Tensor, group=None, wire_dtype: Optional[torch.dtype] = None):
    """SUM all-reduce started on the backend's own stream (RCCL: overlaps with kernels issued afterwards on the compute stream);
    ``.wait()`` on the handle orders the compute stream behind it.  ``wire_dtype=torch.bfloat16``: see ``_Bf16Exchange``."""
    if not is_dist():
        return _Done()
    if wire_dtype is torch.bfloat16 and t.dtype is not torch.bfloat16:
        return _Bf16Exchange(t, group)
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)


def global_count(local: int, device, group=None) -> int:
    if not is_dist():
        return int(local)
    t = torch.tensor([float(local)], device=device, dtype=torch.float64)
    dist.all_reduce(t, group=group)
    return int(round(t.item()))


def global_counts(local, device, group=None):
    """Element-wise global sums of a list of local counts in ONE all-reduce (the same collective on every rank, whatever its shard)."""
    if not is_dist():
        return [int(v) for v in local]
    t = torch.tensor([float(v) for v in local], device=device, dtype=torch.float64)
    dist.all_reduce(t, group=group)
    return [int(round(v)) for v in t.tolist()]


def mean_episode_cost(sum_cost: float, n_episodes: float, device, group=None) -> Tuple[float, float]:
    """Global Jc = sum of finished-episode costs / number of finished episodes (identical on every rank)."""
    t = torch.tensor([float(sum_cost), float(n_episodes)], device=device, dtype=torch.float64)
    allreduce_sum_(t, group)
    s, n = t.tolist()
    return (s / n if n > 0 else 0.0), n


def preflight(model, device) -> dict:
    """Before anything is timed on N > 1 ranks: the collectives the update will issue, on sentinels, checked on EVERY rank --
    one asynchronous SUM all-reduce per tower range of the flat gradient arena (the same tensors, sizes and call as engine.py's per-tower
    exchange; rank r contributes r + 1, so every element must come back as N (N + 1) / 2), the [sum cost, #episodes] reduction, and the
    backend's own idea of the world size.  Raises on any mismatch; returns what it saw (bench.py puts it in the line).  The gradient buffer is
    zero on return."""
    if not is_dist():
        return {"ranks": 1, "backend": None, "tower_ranges_checked": 0}
    n, r = dist.get_world_size(), dist.get_rank()
    want = n * (n + 1) / 2.0
    ar = model.arena
    handles = []
    for lo, hi in ar.tower_ranges:
        ar.flat_g[lo:hi].fill_(float(r + 1))
        handles.append(allreduce_sum_async(ar.flat_g[lo:hi]))
    for h in handles:
        h.wait()
    bad = 0
    for lo, hi in ar.tower_ranges:
        g = ar.flat_g[lo:hi]
        bad += int((g != want).sum().item())
        g.zero_()
    jc, ne = mean_episode_cost(float(r + 1), 1.0, device)
    cnt = global_count(1, device)
    if bad or cnt != n or abs(ne - n) > 0 or abs(jc - want / n) > 1e-12:
        raise RuntimeError(f"collective pre-flight failed on rank {r}: {bad} gradient elements != {want}, rank count {cnt} (world {n}), cost reduction ({jc}, {ne})")
    return {"ranks": cnt, "backend": dist.get_backend(), "tower_ranges_checked": len(ar.tower_ranges),
            "bytes_per_tower_allreduce": [int((hi - lo) * 4) for lo, hi in ar.tower_ranges]}


def gather_floats(v: float, device) -> List[float]:
    """the value of every rank, on every rank (per-rank step times of bench.py)"""
    if not is_dist():
        return [float(v)]
    t = torch.zeros(dist.get_world_size(), device=device, dtype=torch.float64)
    t[dist.get_rank()] = float(v)
    dist.all_reduce(t)
    return t.tolist()


def barrier():
    if is_dist():
        dist.barrier()
