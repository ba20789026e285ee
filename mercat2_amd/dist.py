"""Multi-GPU merge of per-rank count tables (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

Replaces the reference's cross-worker merge -- ``ray.get`` of every chunk's dict and the dict
sum in run_mercat2 (bin/mercat2.py:121-127).  Chunks are dealt to ranks round-robin; every rank
filters its own chunks (the per-chunk min_count rule) and accumulates the survivors into its
running table.  The only exchange step is this one: rows are re-partitioned by KEY RANGE
(owner = floor(key * world / 2^bits) of the key's first word), sent with ONE all_to_all straight
between peers -- key words and count of a row travel side by side; each pair of GPUs has its own
xGMI link, so all 7 links carry traffic at once; no ring -- and insert-added at the owner.  Each
owner then holds a contiguous, sorted key range, so the globally sorted table is the concatenation of
the ranks' exports in rank order.  Two-word keys (33..64-mers) travel the same way as {hi, lo, count}.

A sample that is ONE chunk (file below the -s threshold) has one filter unit: ``count_single_chunk``
splits its records into one range per rank, counts them unfiltered, merges, and applies min_count at
the owners after the merge (lib/mercat2_kmers.py:73-76 applied once per file).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

_SIGN = -(1 << 63)

# figures of this rank's last merge_ranks call (for bench.py's per-rank lines): rows and bytes that left for / arrived
# from OTHER ranks, and the wall seconds of the three phases
LAST_MERGE: dict = {}


def _ordered(keys: torch.Tensor) -> torch.Tensor:
    """uint64 keys stored in int64 tensors: flip the sign bit so signed order == unsigned order."""
    return keys ^ _SIGN


def range_bounds(key_bits: int, world: int) -> List[int]:
    """First key owned by rank 1..world-1 when [0, 2^key_bits) is cut into `world` equal ranges."""
    return [((i << key_bits) + world - 1) // world for i in range(1, world)]


def split_points(sorted_keys: torch.Tensor, key_bits: int, world: int) -> torch.Tensor:
    """Index of the first row of every rank's range in an ascending (unsigned) key tensor."""
    b = torch.tensor([x - (1 << 64) if x >= (1 << 63) else x for x in range_bounds(key_bits, world)],
                     dtype=torch.int64, device=sorted_keys.device)
    cut = torch.searchsorted(_ordered(sorted_keys).contiguous(), _ordered(b), right=False)
    zero = torch.zeros(1, dtype=cut.dtype, device=cut.device)
    end = torch.full((1,), sorted_keys.numel(), dtype=cut.dtype, device=cut.device)
    return torch.cat([zero, cut, end])


def exchange_rows(rows: torch.Tensor, key_bits: int, extra: int = 0, group=None, always: bool = False) -> Tuple[torch.Tensor, List[int]]:
    """All-to-all of table rows by key range.  ``rows`` is (n, w) int64: the key's word(s) then the
    count, ascending by key (unsigned; rows[:, 0] is the most significant word and ``key_bits`` the bits
    it uses).  Returns (the rows this rank owns = what every peer sent, in peer order; every rank's
    ``extra``).  Two collectives: the row counts (with ``extra`` riding along) and the rows."""
    world = dist.get_world_size(group)
    if world == 1 and not always:  # (always: run the collectives even for one rank -- a test of the RCCL path on one GPU)
        return rows, [int(extra)]
    dev = rows.device
    if dev.type != "cpu" and dist.get_backend(group) == "gloo":
        # gloo moves host memory: stage through the CPU (test / single-GPU rehearsal path; with
        # backend "nccl" = RCCL the tensors stay on the device and travel over xGMI)
        got, extras = exchange_rows(rows.cpu(), key_bits, extra, group, always)
        return got.to(dev), extras
    pts = split_points(rows[:, 0], key_bits, world)
    send = (pts[1:] - pts[:-1]).to(torch.int64)
    meta = torch.stack([send, torch.full_like(send, int(extra))], dim=1).contiguous()
    got_meta = torch.empty_like(meta)
    dist.all_to_all_single(got_meta, meta, group=group)
    got_meta = got_meta.cpu()
    send_l, recv_l = send.tolist(), got_meta[:, 0].tolist()
    out = torch.empty((int(sum(recv_l)), rows.shape[1]), dtype=rows.dtype, device=dev)
    dist.all_to_all_single(out, rows.contiguous(), recv_l, send_l, group=group)
    return out, [int(x) for x in got_meta[:, 1].tolist()]


def exchange_pairs(keys: torch.Tensor, counts: torch.Tensor, key_bits: int, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """One-word rows given as two arrays (kept for callers that hold them that way)."""
    rows, _ = exchange_rows(torch.stack([keys, counts], dim=1), key_bits, 0, group)
    return rows[:, 0].contiguous(), rows[:, 1].contiguous()


def balanced_bounds(ctx, key_bits: int, world: int, group=None, device=None, per_rank: int = 2048) -> List[int]:
    """Owner bounds with about equal ROWS per owner (SURVEY 8e "sampled splitters"): every rank samples about
    ``per_rank`` of its keys (mk_sample_keys: every stride-th slot of the hashed table), the samples are all-gathered
    and cut at their quantiles.  Every rank computes the same bounds.  Real tables are far from uniform over the key
    space (canonical keys start with A or C, GC skew, poly-A): with equal key ranges the slowest owner sets the step."""
    rows = max(1, ctx.rows())
    mine = ctx.sample_keys(max(1, rows // per_rank), cap=8 * per_rank)
    dev = device if device is not None else torch.device("cuda", ctx.device)
    if dev.type != "cpu" and dist.get_backend(group) == "gloo":
        dev = torch.device("cpu")
    cap = 8 * per_rank
    buf = torch.zeros(cap + 1, dtype=torch.int64, device=dev)
    buf[0] = int(mine.size)
    if mine.size:
        buf[1:1 + mine.size] = torch.from_numpy(mine.view(np.int64)).to(dev)
    got = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(got, buf, group=group)
    parts = [g[1:1 + int(g[0].item())].cpu().numpy().view(np.uint64) for g in got]
    allk = np.sort(np.concatenate(parts)) if parts else np.zeros(0, np.uint64)
    if allk.size < 8 * world:
        return range_bounds(key_bits, world)
    return [int(allk[allk.size * i // world]) for i in range(1, world)]


def merge_ranks(ctx, key_bits: int, group=None, device=None, min_count: int = 0, always: bool = False,
                balanced: bool = False) -> int:
    """Re-partition ctx's running table across the ranks of `group` by key range and sum.
    On return ctx holds exactly the packed rows of its own range; rows kept as text (characters
    outside the alphabet, k > 64: rare) all sit on rank 0.  With ``min_count`` > 1 rows whose merged
    count is below it are dropped at their owner (single-chunk samples: the filter comes after the
    merge).  Returns the number of rows this rank now owns.

    The rows are grouped by owner ON THE DEVICE in one histogram pass and one scatter pass over the table
    (mk_bucket_rows_device: interleaved {key word(s), count} rows, owner after owner -- no sort, no torch.cat), sent
    with one all_to_all_single, and insert-added at the owner (mk_import_rows_device).  ``balanced``: owner bounds
    from sampled keys instead of equal key ranges.  Dense tables (k * bits <= 15) are ONE reduce of the bins to rank 0."""
    import time
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1 and not always:
        if min_count > 1:
            ctx.filter_min(min_count)
        return ctx.rows()
    t_begin = time.perf_counter()
    t_bucket = t_wire = t_begin
    wire_out = wire_in = rows_out = rows_in = 0
    dev = device if device is not None else torch.device("cuda", ctx.device)
    staged = dev.type != "cpu" and dist.get_backend(group) == "gloo"  # gloo moves host memory (test / rehearsal path)
    mode = ctx.stats()["mode_name"]
    words = ctx.words_per_key()
    ex_k, ex_c = ctx.export_exotic()  # (nothing but a size query when the context holds no text rows)
    if mode == "dense":
        nbins = 1 << ((2 if ctx.alphabet == 0 else 5) * ctx.k)
        bins = torch.empty(nbins, dtype=torch.int64, device=dev)
        ctx.dense_bins_device(bins.data_ptr(), nbins, False)
        if staged:
            host = bins.cpu()
            dist.reduce(host, dst=0, op=dist.ReduceOp.SUM, group=group)
            bins.copy_(host)
        else:
            dist.reduce(bins, dst=0, op=dist.ReduceOp.SUM, group=group)
        _wait_current_stream(dev)
        ctx.reset()
        if rank == 0:
            ctx.dense_bins_device(bins.data_ptr(), nbins, True)
        extras = _all_extras(int(ex_c.size), world, dev if not staged else torch.device("cpu"), group)
    else:
        packed = mode in ("hash64", "hash128")  # byref keeps every row as text
        # (first word of a two-word key: nucleotides fill it; an amino-acid key is a number of 5 k bits)
        bits = (max(1, 5 * ctx.k - 64) if ctx.alphabet == 1 else 64) if words == 2 else key_bits
        bounds = (balanced_bounds(ctx, bits, world, group, dev) if (balanced and packed) else range_bounds(bits, world))
        cap = ctx.rows() + 1
        rows = torch.empty((cap, words + 1), dtype=torch.int64, device=dev)
        send_l = ctx.bucket_rows_device(bounds, rows.data_ptr(), cap) if packed else [0] * world
        t_bucket = time.perf_counter()
        meta_dev = torch.device("cpu") if staged else dev
        meta = torch.tensor([[n, int(ex_c.size)] for n in send_l], dtype=torch.int64, device=meta_dev)
        got_meta = torch.empty_like(meta)
        dist.all_to_all_single(got_meta, meta, group=group)
        got_meta = got_meta.cpu()
        recv_l = got_meta[:, 0].tolist()
        extras = [int(x) for x in got_meta[:, 1].tolist()]
        n_send, n_recv = int(sum(send_l)), int(sum(recv_l))
        ctx.reset(n_recv + 1)  # (emptied and sized for this rank's share: a table that fits the caches imports faster)
        if staged:
            out_h = torch.empty((n_recv, words + 1), dtype=torch.int64)
            dist.all_to_all_single(out_h, rows[:n_send].cpu(), recv_l, send_l, group=group)
            out = out_h.to(dev)
        else:
            out = torch.empty((n_recv, words + 1), dtype=torch.int64, device=dev)
            dist.all_to_all_single(out, rows[:n_send], recv_l, send_l, group=group)
        if n_recv:
            _wait_current_stream(dev)  # the rows have arrived before the engine's own stream reads them
        t_wire = time.perf_counter()
        rows_out, rows_in = n_send - int(send_l[rank]), n_recv - int(recv_l[rank])
        wire_out, wire_in = rows_out * 8 * (words + 1), rows_in * 8 * (words + 1)
        if n_recv:
            ctx.import_rows_device(out.data_ptr(), n_recv)
    # rows kept as text: gathered (as objects, through the host) to rank 0 only when some rank has any
    if any(extras):
        gathered = [None] * world if rank == 0 else None
        if dev.type == "cuda":
            with torch.cuda.device(dev):  # (object collectives over RCCL stage through the CURRENT device)
                dist.gather_object((ex_k, ex_c), gathered, dst=0, group=group)
        else:
            dist.gather_object((ex_k, ex_c), gathered, dst=0, group=group)
        if rank == 0:
            for k_arr, c_arr in gathered:
                ctx.import_exotic(k_arr, c_arr)
    if min_count > 1:
        ctx.filter_min(min_count)
    owned = ctx.rows()
    t_end = time.perf_counter()
    LAST_MERGE.clear()
    LAST_MERGE.update(rows_sent=rows_out, rows_received=rows_in, wire_bytes_sent=wire_out, wire_bytes_received=wire_in,
                      bucket_s=t_bucket - t_begin, collectives_s=max(0.0, t_wire - t_bucket), import_s=max(0.0, t_end - t_wire),
                      total_s=t_end - t_begin, rows_owned=owned)
    return owned


def _wait_current_stream(dev) -> None:
    """Host-wait for torch's current stream on ``dev`` only (the collective just issued), not for the whole device."""
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()


def _all_extras(extra: int, world: int, dev, group) -> List[int]:
    t = torch.tensor([extra], dtype=torch.int64, device=dev)
    got = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(got, t, group=group)
    return [int(g.item()) for g in got]


def record_ranges(text, parts: int) -> List[Tuple[int, int]]:
    """Cut FASTA bytes into ``parts`` byte ranges of about equal size that start at record starts
    (a line whose first character is '>'), so that no window spans a cut."""
    mv = memoryview(text)
    n = len(mv)
    data = np.frombuffer(mv, dtype=np.uint8) if n else np.zeros(0, np.uint8)
    cuts = [0]
    for i in range(1, parts):
        at = max(cuts[-1], n * i // parts)
        while at < n:
            nl = np.flatnonzero(data[at:min(n, at + (1 << 20))] == 0x0A)
            hit = [int(at + j + 1) for j in nl.tolist() if at + j + 1 < n and data[at + j + 1] == 0x3E]
            if hit:
                at = hit[0]
                break
            at = min(n, at + (1 << 20))
        cuts.append(min(at, n))
    cuts.append(n)
    return list(zip(cuts[:-1], cuts[1:]))


def count_single_chunk(ctx, text, min_count: int, key_bits: int, group=None, device=None) -> int:
    """A sample that is one chunk, split over the ranks: rank r counts the r-th
    record range of ``text`` (every rank holds the same bytes) WITHOUT a filter, the ranks merge, and
    min_count is applied to the merged counts -- the reference applies it once per file
    (lib/mercat2_kmers.py:73-76).  Text in front of the first header belongs to the first range (the
    reference counts it as a record).  Returns the rows this rank owns afterwards."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    a, b = record_ranges(text, world)[rank]
    ctx.reset()
    ctx.count_chunk(memoryview(text)[a:b], 0 if world > 1 else min_count)
    if world == 1:
        return ctx.rows()
    return merge_ranks(ctx, key_bits, group=group, device=device, min_count=min_count)
