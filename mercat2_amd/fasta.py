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


def _write_clean_gz(out_fasta: Path, cleaned, timings: Optional[dict] = None) -> int:
    """<base>_clean.fna.gz exactly as the reference's text-mode gzip writer leaves it (level 9, one flush on close):
    its SIZE decides whether the sample is chunked (bin/mercat2.py:101).  Returns that size."""
    import timeit
    t0 = timeit.default_timer()
    with gzip.open(out_fasta, "wb") as writer:
        writer.write(cleaned)
        writer.flush()  # the reference's text-mode writer flushes once when it is closed: one sync-flush marker in the stream
    if timings is not None:
        timings["gzip_s"] = timeit.default_timer() - t0
    return os.stat(out_fasta).st_size


def removeN_background(fasta: Path, raw, outpath: Path, toupper: bool, writers, timings: Optional[dict] = None):
    """The whole of removeN -- rewrite and ``<base>_clean.fna.gz`` -- on the executor ``writers``, from the file's bytes
    ``raw`` already in memory: for samples whose table the GPU takes from the raw text (harness.run_raw_clean), so that
    nothing of removeN stands in front of the first kernel.  Returns ``(path, future, holder)``; ``future.result()`` is
    ``(size of the finished .gz, stats)``; the cleaned bytes are left in ``holder["text"]`` unless the caller has
    called ``holder["drop"]()`` (it did not need them: no reason to keep a second copy of the sample in memory)."""
    import threading
    os.makedirs(outpath, exist_ok=True)
    basename = Path(fasta).stem.split(".")[0]
    out_fasta = Path(outpath, f"{basename}_clean.fna.gz")
    lock = threading.Lock()
    holder = {"dropped": False}

    def drop():
        with lock:
            holder["dropped"] = True
            holder.pop("text", None)
    holder["drop"] = drop

    def job():
        import timeit
        t0 = timeit.default_timer()
        cleaned, stats = clean_text(raw, toupper)
        if timings is not None:
            timings["clean_s"] = timeit.default_timer() - t0
        with lock:
            if not holder["dropped"]:
                holder["text"] = cleaned
        return _write_clean_gz(out_fasta, cleaned, timings), stats
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
