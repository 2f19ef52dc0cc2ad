"""Multi-GPU driver logic (one process per GPU, torch.distributed; backend "nccl" = RCCL on ROCm,
"gloo" in the CPU tests).

Mode R: the database is replicated, the reads of a batch are sharded over the ranks,
and there is NO collective on the data path.  The only communication is the hand-over of each rank's
top-candidate lists to rank 0, where host-side taxonomy assignment happens (north_star: "per-rank
partial hit lists gathered over RCCL/xGMI before host-side taxonomy assignment").
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n: int, rank: int, world: int) -> tuple[int, int]:
    """contiguous, balanced shard [lo, hi) of n queries for this rank (sizes differ by at most 1)"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_candidates(local: torch.Tensor, dst: int = 0, group=None):
    """local: int32 [m, K, 4] candidates of this rank's shard (m may differ by one between ranks).
    Returns on dst the list of per-rank tensors (in rank order = original read order), else None."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return [local]
    m = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(m) for _ in range(world)]
    dist.all_gather(sizes, m, group=group)
    mmax = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros((mmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.zeros_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return [b[: int(s.item())] for b, s in zip(bufs, sizes)]


def classify_sharded(num_queries: int, classify_fn, dst: int = 0, group=None):
    """classify_fn(lo, hi) -> int32 tensor [hi-lo, K, 4] for queries lo..hi-1 (runs the hot path on
    this rank's GPU).  Returns on dst the concatenated [num_queries, K, 4] tensor, else None."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(num_queries, rank, world)
    local = classify_fn(lo, hi)
    if world == 1:
        return local
    parts = gather_candidates(local, dst=dst, group=group)
    return torch.cat(parts, dim=0) if parts is not None else None


# ------------------------------------------------------------------------------------------------------
# Mode P: a partitioned database, ONE PART PER GPU (mc_config.single_part = rank).
# Every rank runs the hot path on the same batch of reads against its own part; a read's final top-K
# list is the stable merge, by hits descending, of the per-part top-K lists taken in part order --
# exactly what feeding the parts' candidates one after the other into the reference's sorted insert
# (candidate_generation.hpp:193-201) gives at sequence level.  Communication: one all-gather of
# [n, K, 4] int32 per batch (RCCL over xGMI); nothing else of the path crosses GPUs.
# ------------------------------------------------------------------------------------------------------
def merge_part_candidates(per_part: list[torch.Tensor]) -> torch.Tensor:
    """per_part[p]: int32 [n, K, 4] = (tgt, hits, beg, end) of part p, unused entries hits == 0.
    Returns the merged [n, K, 4]."""
    K = per_part[0].shape[1]
    allc = torch.cat(per_part, dim=1)                                     # [n, P*K, 4] in part order
    hits = allc[:, :, 1].to(torch.int64)
    order = torch.sort(hits, dim=1, descending=True, stable=True).indices[:, :K]
    out = torch.gather(allc, 1, order[:, :, None].expand(-1, -1, 4))
    empty = out[:, :, 1] == 0
    out = out.clone()
    out[:, :, 0][empty] = -1                                              # unused entries: tgt = 0xFFFFFFFF
    out[:, :, 2][empty] = 0
    out[:, :, 3][empty] = 0
    return out


def classify_partitioned(local: torch.Tensor, group=None) -> torch.Tensor:
    """local: this rank's (= this part's) candidates [n, K, 4] for the WHOLE batch.  Every rank returns the
    merged result (all-gather)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    bufs = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(bufs, local.contiguous(), group=group)
    return merge_part_candidates(bufs)
