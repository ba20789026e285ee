"""Drop-in for the count harness of bin/mercat2.py (lines 86-137): ``chunk_files``,
``countKmers``, ``run_mercat2``, plus ``run_sample`` which does the same job without chunk
files (virtual chunking straight from memory to the GPU).

Semantics kept from the reference:
* the min_count filter is applied to every chunk on its own, before the merge (T2);
* chunk iff the ON-DISK size is >= chunk_size MiB (T3);
* no TSV file and a ``None`` path when nothing survives (T7);
* the two prints ``Significant k-mers: N`` / ``No significant k-mers found``.
"""
from __future__ import annotations

import atexit
import os
import threading
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

from . import native
from .chunker import Chunker
from .kmers import find_kmers, guess_alphabet, read_fasta_bytes, read_head


def chunk_files(name: str, filename: str, chunk_size: int, outpath: str) -> Tuple[str, List[str]]:
    """bin/mercat2.py:86-106: split ``filename`` into chunk files iff it is >= chunk_size MiB."""
    if os.stat(filename).st_size >= (chunk_size * 1024 * 1024):
        os.makedirs(outpath, exist_ok=True)
        all_chunks = Chunker(filename, outpath, str(chunk_size) + "M", ">").files
    else:
        all_chunks = [filename]
    return (name, all_chunks)


def countKmers(file, kmer: int, min_count: int, device: int = 0):
    """bin/mercat2.py:112-114."""
    return find_kmers(Path(file), kmer, min_count, device=device)


# Every sample pays fixed costs when it starts from nothing: creating an engine context (stream, pinned and device
# allocations) takes ~3 ms, the pinned block ring of mk_count_file ~5 ms, and a large sample's working buffers (~3 GB
# per context for 100 MiB chunks) are allocated chunk by chunk and freed -- each hipFree a device-wide wait -- when the
# context is closed: 16 of the 132 ms of a 1.6 GB sample (bench.py file_to_tsv, round 4).  Contexts are therefore kept
# (reset, WITH their buffers and their ring) and handed to the next sample of the same shape: up to _POOL_MAX_PER_KEY per
# (k, alphabet, device, canonical), _POOL_MAX in all -- a node has 288 GB per GPU; release_pool() gives the memory back.
_POOL: dict = {}
_POOL_LOCK = threading.Lock()
_POOL_MAX_PER_KEY = 8
_POOL_MAX = 24


def _take_context(k: int, alphabet: int, device: int, canonical: bool) -> native.Counter:
    with _POOL_LOCK:
        idle = _POOL.get((k, alphabet, device, canonical))
        if idle:
            return idle.pop()
    return native.Counter(k, alphabet, device, canonical=canonical)


def _give_back(ctx: native.Counter, key, text_bytes: int) -> None:
    if text_bytes < (1 << 62):  # (1 << 62: the context saw an error, or its sample was not finished)
        try:
            ctx.share_table(None)
            ctx.reset()
            ctx.reset_stats()
            with _POOL_LOCK:
                idle = _POOL.setdefault(key, [])
                if len(idle) < _POOL_MAX_PER_KEY and sum(len(v) for v in _POOL.values()) < _POOL_MAX:
                    idle.append(ctx)
                    return
        except native.MercatHipError:
            pass
    ctx.close()


def release_pool() -> None:
    """Close the idle contexts kept for the next sample (their device and pinned memory is freed)."""
    with _POOL_LOCK:
        for idle in _POOL.values():
            for ctx in idle:
                ctx.close()
        _POOL.clear()


@atexit.register
def _drain_pool() -> None:
    release_pool()


def _device_list(device: int, devices: Optional[Sequence[int]]) -> List[int]:
    """The GPUs a sample may use: ``devices`` when given (several: its chunks are spread over them), else ``device``."""
    if devices:
        return [int(d) for d in devices]
    return [int(device)]


def _take_contexts(kmer: int, alphabet: int, devs: Sequence[int], streams: int, canon: bool, limit: int) -> List[native.Counter]:
    """Contexts in the order mk_count_file wants them (native.plan_contexts): chunk i -> device devs[i mod ndev],
    the chunks of a device taking turns on its ``streams`` contexts; at most ``limit`` contexts (chunks of the sample)."""
    order = native.plan_contexts(list(devs), max(1, int(streams)))[: max(1, limit)]
    ctxs = [_take_context(kmer, alphabet, d, canon) for d in order]
    # the contexts of one GPU upsert their chunks' survivors into ONE running table, the first one's (mk_share_table:
    # one-word hashed tables; the sum at the end of the sample then finds little left to add)
    first = {}
    for c in ctxs:
        lead = first.setdefault(c.device, c)
        if lead is not c:
            try:
                c.share_table(lead)
            except native.MercatHipError:
                pass  # (other table kinds: every context keeps its own, as before)
    return ctxs


def _sum_into_first(ctxs: Sequence[native.Counter]) -> None:
    """The contexts of one GPU are summed on that GPU, then the GPUs' tables go to ctxs[0] (mk_merge_devices)."""
    leaders = []
    for c in ctxs:
        lead = next((x for x in leaders if x.device == c.device), None)
        if lead is None:
            leaders.append(c)
        else:
            lead.merge_from(c)
            c.reset()
    if len(leaders) > 1:
        native.merge_devices(leaders, native.MERGE_GATHER)


def _finish(ctx: native.Counter, basename: str, out_file, report=print, timings: Optional[dict] = None) -> Tuple[str, Optional[os.PathLike]]:
    rows = ctx.write_tsv(out_file, basename)
    if timings is not None:
        timings["export"] = ctx.export_stats()
    if rows:
        report(f"Significant k-mers: {rows}")
        return basename, out_file
    report("No significant k-mers found")
    return basename, None


def run_mercat2(basename: str, files: Sequence, out_file, kmer: int, min_count: int, num_cores: int = 1,
                *, device: int = 0) -> Tuple[str, Optional[os.PathLike]]:
    """bin/mercat2.py:115-137: count every file (chunk) with its own min_count filter, sum the
    survivors on the GPU, write the TSV sorted by k-mer.  ``num_cores`` is accepted and unused,
    as in the reference."""
    files = list(files)
    alphabet = guess_alphabet(files[0], read_head(files[0])) if files else native.ALPHABET_NT2
    with native.Counter(kmer, alphabet, device) as ctx:
        for f in files:
            native.count_file([ctx], f, 0, min_count)  # every file is one find_kmers call: filtered on its own
        return _finish(ctx, basename, out_file)


def run_text(basename: str, text, out_file, kmer: int, min_count: int, chunk_mib: int = 100, chunked: bool = False,
             *, device: int = 0, devices: Optional[Sequence[int]] = None, streams: Optional[int] = None, canonical: bool = False,
             report=print, keep: Optional[dict] = None, alphabet: Optional[int] = None,
             timings: Optional[dict] = None) -> Tuple[str, Optional[os.PathLike]]:
    """run_sample for FASTA bytes already in memory (e.g. the text removeN just produced, mercat2_amd.fasta):
    ``chunked`` says whether the reference would have chunked the file these bytes stand for (its on-disk size
    against -s, bin/mercat2.py:101); if so the Chunker's cut rule is applied to the bytes and every chunk is
    counted with its own min_count filter, the chunks dealt to ``streams`` contexts on each of ``devices``
    (chunk i -> device i mod N) and the tables summed into the first context."""
    import timeit
    from concurrent.futures import ThreadPoolExecutor
    from .chunker import chunk_offsets
    mv = memoryview(text)
    if alphabet is None:
        alphabet = guess_alphabet("", bytes(mv[:4096]))
    chunk_bytes = max(0, int(chunk_mib)) * 1024 * 1024
    offs = chunk_offsets(mv, chunk_bytes) if (chunked and chunk_bytes > 0) else [0, len(mv)]
    spans = list(zip(offs[:-1], offs[1:]))
    if streams is None:
        streams = native.default_streams(kmer, alphabet)
    canon = bool(canonical and alphabet == native.ALPHABET_NT2)
    devs = _device_list(device, devices)
    ctxs = _take_contexts(kmer, alphabet, devs, streams, canon, len(spans))
    n = len(ctxs)
    size = 1 << 62
    t0 = timeit.default_timer()
    try:
        def share(i):
            for a, b in spans[i::n]:
                ctxs[i].count_chunk(mv[a:b], min_count)
        if n == 1:
            share(0)
        else:
            with ThreadPoolExecutor(n) as pool:
                list(pool.map(share, range(n)))
            _sum_into_first(ctxs)
        size = len(mv)
        t1 = timeit.default_timer()
        result = _finish(ctxs[0], basename, out_file, report)
        if timings is not None:
            timings.update(count_s=t1 - t0, tsv_s=timeit.default_timer() - t1, chunks=len(spans), contexts=n, devices=len(set(devs[:n])))
        if keep is not None and result[1] is not None:
            ctxs[0].trim()
            keep[basename] = ctxs.pop(0)
        return result
    except BaseException:
        size = 1 << 62
        raise
    finally:
        for c in ctxs:
            _give_back(c, (kmer, alphabet, c.device, canon), size)


def run_raw_clean(basename: str, raw, out_file, kmer: int, min_count: int, toupper: bool, limit_bytes: int = 0,
                  *, device: int = 0, canonical: bool = False, report=print, keep: Optional[dict] = None,
                  timings: Optional[dict] = None) -> Optional[Tuple[str, Optional[os.PathLike]]]:
    """The table of a nucleotide sample straight from its RAW text, counted as removeN would leave it (mk_set_clean:
    N runs cut records, text in front of the first header dropped, ``toupper`` after the cut) -- without waiting for
    the host's rewrite and the gzip of ``<base>_clean.fna.gz``.  Only for a sample that is ONE chunk: ``limit_bytes``
    (the -s size, 0 = never chunked) is compared with an upper bound of the cleaned text's size made of the GPU's own
    figures (every N run adds one header line).  Returns None -- nothing counted, nothing written -- when the sample may
    be chunked or holds text the GPU mode does not reproduce; the caller then counts the text removeN produced."""
    import timeit
    mv = memoryview(raw)
    if limit_bytes and len(mv) + len(mv) // 50 + 4096 >= limit_bytes:
        return None
    canon = bool(canonical)
    key = (kmer, native.ALPHABET_NT2, device, canon)
    ctx = _take_context(*key)
    size = 1 << 62
    t0 = timeit.default_timer()
    try:
        ctx.set_clean(True, toupper)
        try:
            ctx.count_chunk(mv, min_count)
        except native.CleanUnsupported:
            ctx.reset()
            size = len(mv)
            return None
        st = ctx.clean_stats()
        # cleaned size <= raw + (one header line per N run, none longer than all non-sequence bytes together) + re-wrapping
        bound = len(mv) + st["n_runs"] * ((len(mv) - st["symbols"] - st["n_bytes"]) + 16) + len(mv) // 40 + 64
        if limit_bytes and bound >= limit_bytes:
            ctx.reset()
            size = len(mv)
            return None
        t1 = timeit.default_timer()
        result = _finish(ctx, basename, out_file, report)
        size = len(mv)
        if timings is not None:
            timings.update(count_s=t1 - t0, tsv_s=timeit.default_timer() - t1, chunks=1, contexts=1, devices=1, gpu_clean=1,
                           n_runs=st["n_runs"], gc_gpu=round(100.0 * st["gc_count"] / max(1, st["symbols"]), 4))
        if keep is not None and result[1] is not None:
            ctx.set_clean(False)
            ctx.trim()
            keep[basename] = ctx
            ctx = None
        return result
    except BaseException:
        size = 1 << 62
        raise
    finally:
        if ctx is not None:
            try:
                ctx.set_clean(False)
            except native.MercatHipError:
                size = 1 << 62
            _give_back(ctx, key, size)


# a file that is ONE filter unit is only spread over several GPUs from this size on (mk_count_file: MK_SPLIT_MIN)
SPLIT_MIN_BYTES = 64 << 20


def run_sample(basename: str, file, out_file, kmer: int, min_count: int, chunk_mib: int = 100,
               *, device: int = 0, devices: Optional[Sequence[int]] = None, streams: Optional[int] = None, canonical: bool = False,
               threads: int = 0, stats: Optional[dict] = None, report=print, keep: Optional[dict] = None,
               timings: Optional[dict] = None) -> Tuple[str, Optional[os.PathLike]]:
    """chunk_files + run_mercat2 in one step with no chunk files (mk_count_file): native reader
    threads read (inflate) the file once into pinned blocks, the reference's cut rule is applied to
    the stream, and each chunk is copied to the GPU and counted (filtered on its own) while the next
    one is being read.  Same TSV as the two-step path.

    ``streams`` contexts (HIP streams; default native.default_streams) per GPU take the chunks in turn and count
    concurrently; they are summed on the device at the end.  With several ``devices`` chunk i goes to device i mod N
    (the Ray fan-out of bin/mercat2.py:119-120 with GPUs for workers) and the GPUs' tables are summed into the first
    by peer copies (mk_merge_devices); a file that is one filter unit but large is cut at record starts, counted
    unfiltered on all of them and filtered after the sum.  ``threads`` = reader threads for plain files (0: pick).
    ``canonical`` is the opt-in extension of mk_set_canonical (not reference behaviour).  If a dict
    is passed as ``stats`` it receives the mk_file_stats_t fields of the read.  ``report`` receives
    the one line the reference prints per sample.  With a dict as ``keep`` the sample's table stays
    on the GPU (``keep[basename]`` = its Counter, working memory released) when it has rows, for
    ``report.merge_counters``; the caller closes it."""
    chunk_bytes = max(0, int(chunk_mib)) * 1024 * 1024
    chunked = chunk_bytes > 0 and os.stat(file).st_size >= chunk_bytes
    alphabet = guess_alphabet(file, read_head(file))
    if streams is None:
        streams = native.default_streams(kmer, alphabet)
    canon = bool(canonical and alphabet == native.ALPHABET_NT2)
    devs = _device_list(device, devices)
    disk = os.stat(file).st_size
    spread = chunked or (len(devs) > 1 and disk >= SPLIT_MIN_BYTES // (4 if str(file).endswith(".gz") else 1))
    if not spread:
        devs = devs[:1]
    limit = len(devs) * max(1, int(streams)) if spread else 1
    if chunked and chunk_bytes:
        limit = min(limit, max(1, -(-disk * (4 if str(file).endswith(".gz") else 1) // chunk_bytes)))
    ctxs = _take_contexts(kmer, alphabet, devs, streams if spread else 1, canon, limit)
    text_bytes = 1 << 62
    import timeit
    t0 = timeit.default_timer()
    try:
        st = native.count_file(ctxs, file, chunk_bytes, min_count, threads)
        text_bytes = st["text_bytes"]
        if stats is not None:
            stats.update(st)
        t1 = timeit.default_timer()
        result = _finish(ctxs[0], basename, out_file, report, timings)
        if timings is not None:
            timings.update(count_s=t1 - t0, tsv_s=timeit.default_timer() - t1, chunks=st["chunks"], contexts=st["contexts"],
                           devices=st["devices"], merge_s=st["s_merge"], wait_io_s=st["s_wait_io"], split_pieces=st["split_pieces"])
        if keep is not None and result[1] is not None:
            ctxs[0].trim()
            keep[basename] = ctxs.pop(0)
        return result
    except BaseException:
        text_bytes = 1 << 62  # (a context that saw an error is not reused)
        raise
    finally:
        for c in ctxs:
            _give_back(c, (kmer, alphabet, c.device, canon), text_bytes)
