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


def gather_candidates_async(local: torch.Tensor, recv: list[torch.Tensor] | None, dst: int = 0, group=None):
    """Equal shard sizes known in advance (the streaming case: fixed batches per rank): no size exchange, no host wait.
    local: [m, K, 4] on every rank; recv: on dst a list of world tensors like local, elsewhere None.  Returns the
    torch.distributed work handle (wait() before reusing local / reading recv), or None with a single rank.  The copy runs on
    the backend's own stream / thread, i.e. concurrently with the next batch's kernels."""
    if not dist.is_initialized():
        if recv is not None:
            recv[0].copy_(local)
        return None
    return dist.gather(local, recv if dist.get_rank(group) == dst else None, dst=dst, group=group, async_op=True)


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
# [n, K, 4] int32 per batch (RCCL over xGMI); nothing else of the path crosses GPUs.  With taxon merging (-lowest above
# sequence) the candidates' taxon keys travel along and the merge replays the reference's per-taxon insert over the parts' lists.
# ------------------------------------------------------------------------------------------------------
def merge_part_candidates(per_part: list[torch.Tensor], per_part_taxa: list[torch.Tensor] | None = None) -> torch.Tensor:
    """per_part[p]: int32 [n, K, 4] = (tgt, hits, beg, end) of part p, unused entries hits == 0.
    per_part_taxa (taxon merging, -lowest above sequence): per_part_taxa[p][n, K] = taxon key of every candidate (0 = none).
    Returns the merged [n, K, 4]."""
    if per_part_taxa is not None:
        return _merge_part_candidates_by_taxon(per_part, per_part_taxa)
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


def _merge_part_candidates_by_taxon(per_part: list[torch.Tensor], per_part_taxa: list[torch.Tensor]) -> torch.Tensor:
    """Taxon merging across parts (candidate_generation.hpp:203-228): at most one entry per taxon, an entry is replaced only by
    strictly more hits and then moves up behind the entries with at least as many.  What that sequential insert leaves is, per
    taxon, its best hit count (first target to reach it, in (part, target) order) and, among equal counts, the order in which the
    entries REACHED their final count.  A part's own list already is the top-K of exactly that ordering over its targets, and a
    taxon outside a part's top-K cannot enter the overall top-K through that part (K taxa of the part are at least as good and as
    early) -- so replaying the insert over the per-part lists, part after part, list order inside a part, gives the result of the
    insert over all targets.  Vectorised over the reads; K * parts steps."""
    K = per_part[0].shape[1]
    n = per_part[0].shape[0]
    dev = per_part[0].device
    allc = torch.cat(per_part, dim=1)
    allt = torch.cat(per_part_taxa, dim=1).to(torch.int64)
    top = torch.zeros((n, K, 4), dtype=allc.dtype, device=dev)
    top[:, :, 0] = -1
    toptax = torch.zeros((n, K), dtype=torch.int64, device=dev)
    ar = torch.arange(K, device=dev)[None, :]
    for c in range(allc.shape[1]):
        cand = allc[:, c, :]
        h = cand[:, 1].to(torch.int64)
        t = allt[:, c]
        th = top[:, :, 1].to(torch.int64)
        live = (h > 0) & (t != 0)
        same = (toptax == t[:, None]) & (th > 0) & live[:, None]
        found = same.any(dim=1)
        pos = torch.where(found, same.to(torch.int64).argmax(dim=1), torch.full((n,), K, device=dev))
        better = found & (h > th.gather(1, pos.clamp(max=K - 1)[:, None])[:, 0])
        fresh = live & ~found
        # where the candidate lands: behind every entry with >= hits (among the entries before its old place / in the whole list)
        ge = th >= h[:, None]
        land_better = (ge & (ar < pos[:, None])).sum(dim=1)
        land_fresh = (ge & (th > 0)).sum(dim=1)
        land = torch.where(better, land_better, land_fresh)
        upto = torch.where(better, pos, torch.full((n,), K - 1, device=dev))        # entries land .. upto-1 move one place down
        act = better | (fresh & (land < K))
        # new list: j < land: old j; j == land: candidate; land < j <= upto: old j - 1; j > upto: old j
        src = torch.where((ar > land[:, None]) & (ar <= upto[:, None]), ar - 1, ar)
        src = torch.where(act[:, None], src, ar)
        ntop = top.gather(1, src[:, :, None].expand(-1, -1, 4))
        ntax = toptax.gather(1, src)
        put = act[:, None] & (ar == land[:, None])
        top = torch.where(put[:, :, None], cand[:, None, :].expand(-1, K, -1), ntop)
        toptax = torch.where(put, t[:, None].expand(-1, K), ntax)
    return top


def classify_partitioned(local: torch.Tensor, group=None, taxa: torch.Tensor | None = None) -> torch.Tensor:
    """local: this rank's (= this part's) candidates [n, K, 4] for the WHOLE batch; taxa: with taxon merging (-lowest above
    sequence) the taxon key [n, K] of every candidate (0 = none), else None.  Every rank returns the merged result (all-gather)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    bufs = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(bufs, local.contiguous(), group=group)
    if taxa is None:
        return merge_part_candidates(bufs)
    tb = [torch.zeros_like(taxa) for _ in range(world)]
    dist.all_gather(tb, taxa.contiguous(), group=group)
    return merge_part_candidates(bufs, tb)


# ------------------------------------------------------------------------------------------------------
# Mode K: ONE database part whose features are key-sharded over the GPUs
# (mc_config.key_shard_index / key_shard_count; every feature lives on exactly one GPU).  Per batch:
#   1. every rank runs the hot path up to the sorted location lists on ALL reads of the batch against ITS keys
#      (mc_query_device(MC_WANT_ALLHITS) on a sharded context) -> partial lists;
#   2. the partial lists travel to the rank that owns the read (contiguous read shards, shard_bounds):
#      one all-to-all of the per-read counts, one all-to-all-v of the locations (RCCL over xGMI);
#   3. the owner concatenates the partial lists of each of its reads and runs rows 8-10 on the union
#      (mc_candidates_from_hits) -- a full sort of the union = the single-part reference result, bit for bit.
# (The sketches are recomputed on every rank instead of all-gathered: 0.28 ms per 10^6 reads of ALU work per rank
#  against 128 MB of traffic per 10^6 reads; revisit when the sketch time shows up in the 8-GPU scaling.)
# ------------------------------------------------------------------------------------------------------
def exchange_partial_hits(counts: torch.Tensor, hits: torch.Tensor, group=None):
    """counts: int64 [n] locations per read in this rank's partial lists; hits: int64 [sum(counts)] the lists back to back
    in read order.  Returns for this rank's read shard [lo, hi): (per_source_counts int64 [world, hi-lo],
    per_source_hits list of int64 tensors in source-rank order)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = counts.numel()
    if world == 1:
        return counts[None, :], [hits]
    bounds = [shard_bounds(n, r, world) for r in range(world)]
    lo, hi = bounds[rank]
    send_reads = [b[1] - b[0] for b in bounds]
    recv_counts = torch.empty(world * (hi - lo), dtype=torch.int64, device=counts.device)
    dist.all_to_all_single(recv_counts, counts.contiguous(), output_split_sizes=[hi - lo] * world, input_split_sizes=send_reads, group=group)
    recv_counts = recv_counts.view(world, hi - lo)
    offs = torch.zeros(n + 1, dtype=torch.int64, device=counts.device)
    offs[1:] = torch.cumsum(counts, 0)
    send_elems = [int(offs[b[1]] - offs[b[0]]) for b in bounds]
    recv_elems = [int(x) for x in recv_counts.sum(dim=1).tolist()]
    recv_hits = torch.empty(sum(recv_elems), dtype=hits.dtype, device=hits.device)
    dist.all_to_all_single(recv_hits, hits.contiguous(), output_split_sizes=recv_elems, input_split_sizes=send_elems, group=group)
    return recv_counts, list(torch.split(recv_hits, recv_elems))


def union_partial_hits(per_source_counts: torch.Tensor, per_source_hits: list[torch.Tensor]):
    """Concatenates, read by read, the partial lists of all sources (source order inside a read; the order does not matter,
    the union is sorted afterwards).  -> (hit_offsets int64 [m + 1], hits int64 [total])"""
    world, m = per_source_counts.shape
    dev = per_source_counts.device
    total_per_read = per_source_counts.sum(dim=0)
    offsets = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(total_per_read, 0)
    out = torch.empty(int(offsets[-1]), dtype=per_source_hits[0].dtype if per_source_hits else torch.int64, device=dev)
    before = torch.zeros(m, dtype=torch.int64, device=dev)               # locations of earlier sources inside each read
    for s in range(world):
        c = per_source_counts[s]
        ne = int(c.sum())
        if ne:
            src_off = torch.cumsum(c, 0) - c                             # start of each read inside this source's block
            read_of = torch.repeat_interleave(torch.arange(m, device=dev), c)
            within = torch.arange(ne, device=dev) - src_off[read_of]
            out[offsets[:-1][read_of] + before[read_of] + within] = per_source_hits[s]
        before = before + c
    return offsets, out


def classify_key_sharded(counts: torch.Tensor, hits: torch.Tensor, candidates_fn, group=None):
    """counts/hits: this rank's partial lists for the WHOLE batch (step 1).  candidates_fn(hit_offsets, hits) -> int32
    [m, K, 4] runs rows 8-10 for this rank's read shard (step 3).  Returns this rank's [m, K, 4] (gather_candidates hands
    them to rank 0)."""
    psc, psh = exchange_partial_hits(counts, hits, group=group)
    offsets, union = union_partial_hits(psc, psh)
    return candidates_fn(offsets, union)


# ---- Mode K, device data path ----------------------------------------------------------------------------------------------
# The functions above are the reference semantics of the exchange in torch (and what the gloo tests run).  On GPUs the per-batch
# path is: ONE small device-to-host copy (the world + 1 list offsets that become the all-to-all-v split sizes), one tiny
# all-to-all of those sizes, the all-to-all of the per-read counts (int32) and of the locations -- and then
# mc_candidates_from_partial_hits: union of the sources' pieces per read (union_copy_kernel), sort, window ranges, top candidates,
# all on the device, no per-rank host round trips and no torch index arithmetic per source.
def partial_lists_of(db, res, n: int, device):
    """the partial location lists a key-sharded context returned for a batch (mc_query_device(MC_WANT_ALLHITS)), as torch tensors:
    (offsets int64 [n + 1], hits int64 [total])"""
    off = torch.empty(n + 1, dtype=torch.int64, device=device)
    db.copy_results(off.data_ptr(), res.hit_offsets, (n + 1) * 8)
    db.synchronize()
    total = int(off[-1])                                                   # sizes the hits tensor
    hits = torch.empty(max(total, 1), dtype=torch.int64, device=device)
    if total:
        db.copy_results(hits.data_ptr(), res.hits, total * 8)
        db.synchronize()
    return off, hits[:total]


def exchange_partial_lists(offsets: torch.Tensor, hits: torch.Tensor, group=None):
    """offsets int64 [n + 1] / hits int64: this rank's partial lists for the WHOLE batch.  Sends every rank the pieces of ITS reads
    (contiguous read shards, shard_bounds).  -> (counts int32 [world * m] source-major, hits int64 [total], total) for this rank's
    m reads: the inputs of mc_candidates_from_partial_hits."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = offsets.numel() - 1
    counts = (offsets[1:] - offsets[:-1]).to(torch.int32)
    if world == 1:
        return counts, hits, int(hits.numel())
    bounds = [shard_bounds(n, r, world) for r in range(world)]
    lo, hi = bounds[rank]
    m = hi - lo
    cut = torch.tensor([b[0] for b in bounds] + [n], dtype=torch.int64, device=offsets.device)
    send_elems_t = offsets[cut][1:] - offsets[cut][:-1]                   # locations going to each rank
    recv_elems_t = torch.empty_like(send_elems_t)
    dist.all_to_all_single(recv_elems_t, send_elems_t, group=group)
    sizes = torch.stack([send_elems_t, recv_elems_t]).cpu()               # the one host round trip of the exchange
    send_elems, recv_elems = [int(x) for x in sizes[0]], [int(x) for x in sizes[1]]
    recv_counts = torch.empty(world * m, dtype=torch.int32, device=offsets.device)
    dist.all_to_all_single(recv_counts, counts.contiguous(), output_split_sizes=[m] * world, input_split_sizes=[b[1] - b[0] for b in bounds], group=group)
    total = sum(recv_elems)
    recv_hits = torch.empty(max(total, 1), dtype=hits.dtype, device=hits.device)
    dist.all_to_all_single(recv_hits[:total], hits.contiguous(), output_split_sizes=recv_elems, input_split_sizes=send_elems, group=group)
    return recv_counts, recv_hits[:total], total


def classify_key_sharded_device(db, res, n: int, K: int, max_win_uniform: int, lowest: int = 0, group=None, wire: int = 4) -> torch.Tensor:
    """Mode K for one batch on this rank's GPU: res = db.query_device(..., want_partial_hits=True) (or want_allhits) of the key-sharded
    context db over all n reads.  Returns int32 [m, K, 4] for this rank's read shard (gather_candidates hands the shards to rank 0).
    wire = 4 (default): the lists travel as 4-byte global window numbers (mc_partial_numbers / mc_candidates_from_partial_numbers; falls
    back to 8 when the table has no such numbering); wire = 8: as (target, window) pairs (mc_candidates_from_partial_hits)."""
    if wire == 4 and db.table_layout()["location_bytes"] == 4:
        return classify_key_sharded_numbers(db, res, n, K, max_win_uniform, lowest=lowest, group=group)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    device = torch.device("cuda", db.cfg.device)
    off, hits = partial_lists_of(db, res, n, device)
    counts, rhits, total = exchange_partial_lists(off, hits, group=group)
    # The all-to-alls run on torch's current stream (RCCL only makes THAT stream wait, the host is not blocked); the owner-side
    # kernels run on the context's own non-blocking stream.  Order them: the receive buffers must be complete before union_totals /
    # union_copy read them.
    if device.type == "cuda" and torch.cuda.is_available():
        torch.cuda.current_stream(device).synchronize()
    lo, hi = shard_bounds(n, rank, world)
    m = hi - lo
    r2 = db.candidates_from_partial_hits(counts.data_ptr(), rhits.data_ptr() if total else 0, total, m, world, max_win_uniform=max_win_uniform, lowest=lowest)
    out = torch.empty((m, K, 4), dtype=torch.int32, device=device)
    db.copy_results(out.data_ptr(), r2.cands, m * K * 16)
    db.synchronize()
    return out


def exchange_numbers(counts: torch.Tensor, numbers: torch.Tensor, cut_offsets, n: int, group=None):
    """The all-to-all-v of Mode K with 4-byte locations.  counts int32 [n] / numbers int32 [total]: this rank's partial lists for the
    WHOLE batch (numbers back to back in read order); cut_offsets (host, [world + 1]): where each rank's read shard begins in numbers --
    known on the host without another device round trip (mc_partial_numbers).  -> (recv_counts int32 [world * m] source-major,
    recv_numbers int32 [total_recv + 4], source_offsets list [world + 1]) for this rank's m reads."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    send_elems = [int(cut_offsets[r + 1]) - int(cut_offsets[r]) for r in range(world)]
    if world == 1:
        return counts, numbers, [0, send_elems[0]]                           # (the caller's buffer has the 16 bytes of slack the owner's kernels need)
    bounds = [shard_bounds(n, r, world) for r in range(world)]
    m = bounds[rank][1] - bounds[rank][0]
    sizes = torch.tensor(send_elems, dtype=torch.int64, device=counts.device)
    rsizes = torch.empty_like(sizes)
    dist.all_to_all_single(rsizes, sizes, group=group)                       # every rank learns what it will receive
    recv_elems = [int(x) for x in rsizes.cpu()]
    recv_counts = torch.empty(world * m, dtype=counts.dtype, device=counts.device)
    dist.all_to_all_single(recv_counts, counts.contiguous(), output_split_sizes=[m] * world, input_split_sizes=[b[1] - b[0] for b in bounds], group=group)
    total = sum(recv_elems)
    recv_numbers = torch.zeros(total + 4, dtype=numbers.dtype, device=numbers.device)
    dist.all_to_all_single(recv_numbers[:total], numbers.contiguous(), output_split_sizes=recv_elems, input_split_sizes=send_elems, group=group)
    so = [0]
    for e in recv_elems:
        so.append(so[-1] + e)
    return recv_counts, recv_numbers, so


class _DeviceArray:
    """a ctx-owned device buffer as a zero-copy torch tensor (torch.as_tensor reads __cuda_array_interface__): the collectives send the shard's
    numbers from where mc_partial_numbers left them -- copying them into a tensor first cost 2 x 5.1 GB of HBM traffic per 10^6 reads at RefSeq scale"""

    def __init__(self, ptr: int, count: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 2}


def device_view(ptr: int, count: int, dtype: torch.dtype, device: torch.device) -> torch.Tensor:
    ts = {torch.int32: "<i4", torch.int64: "<i8", torch.uint8: "|u1"}[dtype]
    if count == 0 or not ptr:
        return torch.empty(0, dtype=dtype, device=device)
    return torch.as_tensor(_DeviceArray(ptr, count, ts), device=device)


def classify_key_sharded_numbers(db, res, n: int, K: int, max_win_uniform: int, lowest: int = 0, group=None, max_win: torch.Tensor | None = None) -> torch.Tensor:
    """Mode K with 4-byte locations on the wire (the torch.distributed form of what metacache_amd/csrc/keyset.cpp does inside one
    process): shard side mc_partial_numbers, all-to-all-v of counts and numbers, owner side mc_candidates_from_partial_numbers --
    no union copy, the receive buffer is the owner's location store.  max_win: per-read window ranges of the WHOLE batch (int32 [n]) or
    None with max_win_uniform."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    device = torch.device("cuda", db.cfg.device)
    bounds = [shard_bounds(n, r, world) for r in range(world)]
    part, cuts = db.partial_numbers(res, n, [b[0] for b in bounds] + [n])
    db.synchronize()                                                         # the pack kernel ran on the context's stream, the collectives run on torch's
    # zero-copy views of the context's own buffers (valid until the next mc_partial_numbers call; they carry the 16 bytes of slack the owner's
    # kernels need when a single rank hands them straight on)
    counts = device_view(part.counts, n, torch.int32, device)
    numbers = device_view(part.numbers, int(part.total), torch.int32, device)
    rc, rn, so = exchange_numbers(counts, numbers, cuts, n, group=group)
    if device.type == "cuda" and torch.cuda.is_available():
        torch.cuda.current_stream(device).synchronize()                      # the receive buffers are complete before the owner's kernels read them
    lo, hi = bounds[rank]
    m = hi - lo
    mw = max_win[lo:hi].contiguous() if max_win is not None else None
    r2 = db.candidates_from_partial_numbers(rc.data_ptr(), rn.data_ptr(), so, m, max_win_ptr=mw.data_ptr() if mw is not None else 0,
                                            max_win_uniform=0 if mw is not None else max_win_uniform, lowest=lowest)
    out = torch.empty((m, K, 4), dtype=torch.int32, device=device)
    db.copy_results(out.data_ptr(), r2.cands, m * K * 16)
    db.synchronize()
    return out
