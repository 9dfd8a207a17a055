"""Data-parallel sharding of the batch over GPUs: one process per GPU, no data-path collective.

Environments are independent (the map is a constant), so stepping needs ZERO inter-GPU traffic: rank r owns
the contiguous global env ids ``[start_r, start_r + count_r)``.  The Philox counter of an env is its GLOBAL
id (``env_id_base + local index``), so a sharded run produces exactly the episodes a single-GPU run of the
same ids would.  The single collective on the path is the node-wide metrics reduction: every rank sums its
envs' lifetime accumulators on device (``susnet_reduce_lifetime``) and ONE ``all_gather`` (RCCL over xGMI;
``torch.distributed`` backend "nccl" on ROCm) of 12 x int64 = 96 bytes per rank assembles the table --
latency-bound, done once per rollout block, never per step.

The reference has no distributed code at all; this module is new functionality.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Optional, Tuple

import torch
import torch.distributed as dist

from ._lib import LIFETIME_NAMES, N_LIFETIME


def shard_range(global_batch: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous split; the first ``global_batch % world_size`` ranks get one extra env."""
    assert 0 <= rank < world_size and global_batch >= world_size
    base, extra = divmod(global_batch, world_size)
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank); a plain single-process run returns (0, 1, 0) without a group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def make_sharded(factory: Callable[..., object], global_batch: int, rank: int, world_size: int, local_rank: int = 0,
                 **kwargs):
    """``factory(batch=..., env_id_base=..., device=..., **kwargs)`` for this rank's shard."""
    start, count = shard_range(global_batch, rank, world_size)
    return factory(batch=count, env_id_base=start, device=f"cuda:{local_rank}", **kwargs)


def all_gather_totals(local: torch.Tensor, group=None) -> torch.Tensor:
    """ONE all_gather of this rank's int64 totals vector -> ``[world_size, len(local)]`` on every rank."""
    local = local.reshape(-1).contiguous()
    if not (dist.is_available() and dist.is_initialized()):
        return local.unsqueeze(0).clone()
    world = dist.get_world_size(group)
    dev = local.device
    if dist.get_backend(group) != "nccl":  # gloo rehearsals: stage through the host
        local = local.cpu()
    out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local, group=group)
    return out.view(world, local.numel()).to(dev)


def node_metrics(env, group=None) -> Dict[str, int]:
    """Whole-job episode metrics: device reduction per rank + one all-gather + a host sum."""
    table = all_gather_totals(env.lifetime_totals(), group=group)
    tot = table.sum(dim=0).tolist()
    assert len(tot) == N_LIFETIME
    out = {name: int(v) for name, v in zip(LIFETIME_NAMES, tot)}
    out["per_rank_episodes"] = table[:, 0].tolist()
    return out
