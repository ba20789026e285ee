"""Drop-in for the part of MerCat2's ``mercat2_fasta`` module that stands in front of the counting path:
``removeN`` (lib/mercat2_fasta.py:53-119) with its helper ``split_sequenceN`` (:21-49).

The reference runs removeN on every nucleotide FASTA before counting (bin/mercat2.py:239-244, 276) unless
``-skipclean`` is given: sequences are cut at runs of 'N' into ``>{name}_{i} {info}`` records re-wrapped at 80
columns, and the result is written as ``<outpath>/<base>_clean.fna.gz``.  Here the rewrite is native host code
behind the C ABI (``mk_remove_n``, csrc/mk_host.cpp); the cleaned text can be handed to the counting engine
straight from memory (``clean_text``) instead of being read back from the ``.gz``.
"""
from __future__ import annotations

import gzip
import os
from pathlib import Path
from typing import Optional, Dict, Tuple

from . import native
from .kmers import read_fasta_bytes


def clean_text(raw, toupper: bool = False) -> Tuple[bytes, Dict[str, float]]:
    """removeN on FASTA bytes already in memory: (cleaned text, {'GC Content': percent}).  Raises
    ZeroDivisionError for input without sequence and IndexError for a record to be split whose header is
    empty, as the reference does; a byte >= 0x80 in a sequence line raises native.NonAsciiInput (the counting
    engine refuses such text too)."""
    cleaned, st = native.remove_n(raw, toupper)
    return cleaned, {"GC Content": 100.0 * st["gc_count"] / st["total_length"]}


class GzDecision:
    """Whether ``<base>_clean.fna.gz`` reaches ``limit`` bytes -- what decides about chunking (bin/mercat2.py:101) -- as soon as
    that is known: a DEFLATE stream only grows, so the answer is "yes" the moment the bytes written pass the limit, and
    "no" only when the file is complete.  ``wait()`` blocks until then; ``size`` is the bytes written so far."""

    def __init__(self, limit: int):
        import threading
        self.limit = int(limit)
        self.size = 0
        self.chunked: Optional[bool] = None
        self._event = threading.Event()

    def _grew(self, size: int, complete: bool) -> None:
        self.size = size
        if self.chunked is None:
            if self.limit > 0 and size >= self.limit:
                self.chunked = True
                self._event.set()
            elif complete:
                self.chunked = False
                self._event.set()

    def _failed(self) -> None:
        self._event.set()  # (the writer's exception surfaces through its future)

    def wait(self, timeout: Optional[float] = None) -> Optional[bool]:
        self._event.wait(timeout)
        return self.chunked


class _CountingFile:
    """The raw file under GzipFile: counts what reaches it."""

    def __init__(self, path, decision: Optional[GzDecision]):
        self._f = open(path, "wb")
        self._n = 0
        self._d = decision

    def write(self, data):
        n = self._f.write(data)
        self._n += len(data)
        if self._d is not None:
            self._d._grew(self._n, False)
        return n

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def _write_clean_gz(out_fasta: Path, cleaned, timings: Optional[dict] = None, decision: Optional[GzDecision] = None) -> int:
    """<base>_clean.fna.gz exactly as the reference's text-mode gzip writer leaves it (level 9, one flush on close):
    its SIZE decides whether the sample is chunked (bin/mercat2.py:101).  Returns that size.  The text is handed to
    zlib in 1 MiB slices (the stream does not depend on how its input is sliced: tests/test_clean.py), so that
    ``decision`` learns that the limit has been passed while the rest is still being compressed."""
    import timeit
    t0 = timeit.default_timer()
    raw = _CountingFile(out_fasta, decision)
    try:
        # (GzipFile names the member after `filename` and takes the bytes to `fileobj`: the same header gzip.open writes)
        with gzip.GzipFile(filename=os.fspath(out_fasta), mode="wb", fileobj=raw) as writer:
            mv = memoryview(cleaned)
            for a in range(0, len(mv), 1 << 20):
                writer.write(mv[a:a + (1 << 20)])
            writer.flush()  # the reference's text-mode writer flushes once when it is closed: one sync-flush marker in the stream
        raw.close()
    except BaseException:
        raw.close()
        if decision is not None:
            decision._failed()
        raise
    size = os.stat(out_fasta).st_size
    if decision is not None:
        decision._grew(size, True)
    if timings is not None:
        timings["gzip_s"] = timeit.default_timer() - t0
    return size


def removeN_background(fasta: Path, raw, outpath: Path, toupper: bool, writers, timings: Optional[dict] = None, limit: int = 0):
    """The whole of removeN -- rewrite and ``<base>_clean.fna.gz`` -- on the executor ``writers``, from the file's bytes
    ``raw`` already in memory: for samples whose table the GPU takes from the raw text (harness.run_raw_clean), so that
    nothing of removeN stands in front of the first kernel.  Returns ``(path, future, holder)``; ``future.result()`` is
    ``(size of the finished .gz, stats)``; the cleaned bytes are left in ``holder["text"]`` unless the caller has
    called ``holder["drop"]()`` (it did not need them: no reason to keep a second copy of the sample in memory).
    ``holder["ready"]`` is set when the rewrite is done (the text is there, or the job has failed); ``holder["decision"]``
    is the GzDecision of the file against ``limit`` bytes (0: no limit: "not chunked" once the file is complete)."""
    import threading
    os.makedirs(outpath, exist_ok=True)
    basename = Path(fasta).stem.split(".")[0]
    out_fasta = Path(outpath, f"{basename}_clean.fna.gz")
    lock = threading.Lock()
    holder = {"dropped": False, "ready": threading.Event(), "decision": GzDecision(limit)}

    def drop():
        with lock:
            holder["dropped"] = True
            holder.pop("text", None)
    holder["drop"] = drop

    def job():
        import timeit
        t0 = timeit.default_timer()
        try:
            cleaned, stats = clean_text(raw, toupper)
        except BaseException:
            holder["ready"].set()
            holder["decision"]._failed()
            raise
        if timings is not None:
            timings["clean_s"] = timeit.default_timer() - t0
        with lock:
            if not holder["dropped"]:
                holder["text"] = cleaned
        holder["ready"].set()
        return _write_clean_gz(out_fasta, cleaned, timings, holder["decision"]), stats
    return out_fasta.absolute(), writers.submit(job), holder


def removeN_start(fasta: Path, outpath: Path, toupper: bool, writers, timings: Optional[dict] = None):
    """removeN with the level-9 DEFLATE off the critical path: the file is read and rewritten (native code, memory
    speed) here; writing ``<base>_clean.fna.gz`` -- ~1.5 MB/s of zlib on DNA, a hundred times slower than everything
    else on the way to the table -- is handed to the executor ``writers`` (zlib releases the GIL).  Returns
    ``(path, stats, cleaned bytes, future)``; ``future.result()`` is the size of the finished file.  The counting path
    takes the bytes from memory and needs that size only when the cleaned text itself reaches the chunk size."""
    import timeit
    os.makedirs(outpath, exist_ok=True)
    basename = Path(fasta).stem.split(".")[0]
    out_fasta = Path(outpath, f"{basename}_clean.fna.gz")
    t0 = timeit.default_timer()
    raw = read_fasta_bytes(fasta)
    t1 = timeit.default_timer()
    cleaned, stats = clean_text(raw, toupper)
    if timings is not None:
        timings.update(read_s=t1 - t0, clean_s=timeit.default_timer() - t1)
    future = writers.submit(_write_clean_gz, out_fasta, cleaned, timings)
    return out_fasta.absolute(), stats, cleaned, future


def removeN_text(fasta: Path, outpath: Path, toupper: bool, timings: Optional[dict] = None):
    """removeN that also hands back the cleaned text: ``(path, stats, cleaned bytes)`` -- the counting path takes
    the bytes from memory instead of inflating the file it has just written.  ``timings`` receives read_s / clean_s /
    gzip_s (the CLI's -debug line)."""
    import timeit
    os.makedirs(outpath, exist_ok=True)
    basename = Path(fasta).stem.split(".")[0]
    out_fasta = Path(outpath, f"{basename}_clean.fna.gz")
    t0 = timeit.default_timer()
    raw = read_fasta_bytes(fasta)
    t1 = timeit.default_timer()
    cleaned, stats = clean_text(raw, toupper)
    if timings is not None:
        timings.update(read_s=t1 - t0, clean_s=timeit.default_timer() - t1)
    _write_clean_gz(out_fasta, cleaned, timings)
    return out_fasta.absolute(), stats, cleaned


def removeN(fasta: Path, outpath: Path, toupper: bool):
    """Splits sequences in a scaffold fasta file at N repeats (lib/mercat2_fasta.py:53-119: same arguments, same
    ``(path of <base>_clean.fna.gz, {'GC Content': ...})`` result, same file content).  The file is written with
    Python's gzip module exactly as the reference writes it (level 9), so that its size -- which decides whether
    the sample is chunked, bin/mercat2.py:101 -- is the same."""
    path, stats, _ = removeN_text(fasta, outpath, toupper)
    return path, stats
