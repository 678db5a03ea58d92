"""Multi-GPU layout of the rollout: envs are independent, so they are sharded by contiguous id range, one process
per GPU, the packed stream replicated on every rank, and NO collective on the stepping path.  The only exchange is
the per-env episode-return gather (torch.distributed all_gather: RCCL over xGMI with backend "nccl" on GPUs, gloo in
the CPU tests) - 8 bytes per env, latency-bound whatever the algorithm."""
from __future__ import annotations

from typing import Tuple

import torch


def shard_range(n_total: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous env-id range [lo, hi) owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n_total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_episode_returns(local_returns: torch.Tensor) -> torch.Tensor:
    """All ranks' per-env episode returns concatenated in env-id order.  Works for unequal shard sizes."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_returns.clone()
    world = dist.get_world_size()
    n = torch.tensor([local_returns.numel()], dtype=torch.int64, device=local_returns.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    m = int(max(int(s.item()) for s in sizes))
    padded = torch.zeros(m, dtype=local_returns.dtype, device=local_returns.device)
    padded[: local_returns.numel()] = local_returns
    out = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(out, padded)
    return torch.cat([o[: int(s.item())] for o, s in zip(out, sizes)])
