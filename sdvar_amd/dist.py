"""One-process-per-GPU sharding of independent image samples.

The sampling path has no data-path exchange: images are independent, every rank holds full weight replicas and private
KV caches, and the acceptance decision is taken per shard (`accept_scope="shard"`).  The only collective is ONE
all-gather of per-rank counters (accepted tokens, target calls, ...) after the run - RCCL over xGMI on the GPU box
(`backend="nccl"`), gloo in the CPU tests.  Noise is keyed by the GLOBAL image index (sdvar_amd/noise.py), so a shard
produces exactly the tokens the same images would get in a single-process run of the whole batch with shard scope.
The reference has no counterpart (its inference is single-device; dist.py there only serves training).
"""
from __future__ import annotations

import os
from typing import Dict, Sequence, Tuple

import torch
import torch.distributed as tdist

COUNTER_KEYS = ("images", "accepted_tokens", "target_calls", "draft_stage_calls", "forced_accepts")


def init_from_env(device_type: str = "cuda") -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; initialises the process group when world > 1."""
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if device_type == "cuda":
        # rehearsals put several ranks on one GPU (SDVAR_DIST_BACKEND=gloo): map the local rank onto the visible devices
        local = local % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
    if world > 1 and not tdist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SDVAR_DIST_BACKEND") or ("nccl" if device_type == "cuda" else "gloo")   # "nccl" is RCCL on ROCm
        tdist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _coll_device(device):
    """Collectives run on the GPU with RCCL; with the gloo backend (CPU tests, single-GPU rehearsals) on host tensors."""
    if tdist.is_available() and tdist.is_initialized() and tdist.get_backend() == "gloo":
        return "cpu"
    return device


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous batch split: images [lo, hi) of the global batch belong to `rank` (remainder to the low ranks)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_counters(stats: Dict[str, int], device) -> Dict[str, object]:
    """All-gather the per-rank counters; returns totals plus the per-rank table (identical on every rank)."""
    vec = torch.tensor([int(stats.get(k, 0)) for k in COUNTER_KEYS], dtype=torch.int64, device=_coll_device(device))
    if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
        out = [torch.zeros_like(vec) for _ in range(tdist.get_world_size())]
        tdist.all_gather(out, vec)
        table = torch.stack(out).cpu()
    else:
        table = vec.cpu().view(1, -1)
    tot = table.sum(0).tolist()
    res = {k: int(v) for k, v in zip(COUNTER_KEYS, tot)}
    res["per_rank"] = table.tolist()
    res["mean_accepted_tokens_per_step"] = (res["accepted_tokens"] / res["target_calls"]) if res["target_calls"] else 0.0
    return res


def leading_accepted(matched: Sequence[int], totals: Sequence[int], thr: float) -> int:
    """Number of leading stages whose match rate is >= thr, with the reference's arithmetic: float32 mean compared as a
    python float (models/var.py:1203, 1217)."""
    import numpy as np
    n = 0
    for m, t in zip(matched, totals):
        rate = float(np.float32(m) / np.float32(t))
        if rate >= thr:
            n += 1
        else:
            break
    return n


def global_accept(matched: Sequence[int], totals: Sequence[int], thr: float, device) -> Tuple[int, Sequence[int], Sequence[int]]:
    """accept_scope="global" (SURVEY.md section 8e): the reference decides on the mean over the WHOLE batch, so the per-stage
    (matched, total) counts are summed over the ranks - one all-reduce of 2*gamma int64 per round, latency only - and every
    rank takes the same decision.  Returns (n_accept, global matched, global totals)."""
    vec = torch.tensor(list(matched) + list(totals), dtype=torch.int64, device=_coll_device(device))
    if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
        tdist.all_reduce(vec, op=tdist.ReduceOp.SUM)
    v = vec.cpu().tolist()
    g = len(matched)
    return leading_accepted(v[:g], v[g:], thr), v[:g], v[g:]


def max_over_ranks(value: float, device) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=_coll_device(device))
    if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
        tdist.barrier()
