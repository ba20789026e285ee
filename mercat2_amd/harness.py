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

import os
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

from . import native
from .chunker import Chunker, chunk_offsets
from .kmers import find_kmers, guess_alphabet, map_fasta, read_fasta_bytes


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


def _finish(ctx: native.Counter, basename: str, out_file) -> Tuple[str, Optional[os.PathLike]]:
    rows = ctx.write_tsv(out_file, basename)
    if rows:
        print(f"Significant k-mers: {rows}")
        return basename, out_file
    print("No significant k-mers found")
    return basename, None


def run_mercat2(basename: str, files: Sequence, out_file, kmer: int, min_count: int, num_cores: int = 1,
                *, device: int = 0) -> Tuple[str, Optional[os.PathLike]]:
    """bin/mercat2.py:115-137: count every file (chunk) with its own min_count filter, sum the
    survivors on the GPU, write the TSV sorted by k-mer.  ``num_cores`` is accepted and unused,
    as in the reference."""
    files = list(files)
    first = read_fasta_bytes(files[0]) if files else b""
    with native.Counter(kmer, guess_alphabet(files[0] if files else "", first), device) as ctx:
        for i, f in enumerate(files):
            ctx.count_chunk(first if i == 0 else read_fasta_bytes(f), min_count)
        return _finish(ctx, basename, out_file)


def run_sample(basename: str, file, out_file, kmer: int, min_count: int, chunk_mib: int = 100,
               *, device: int = 0, streams: int = 2, canonical: bool = False) -> Tuple[str, Optional[os.PathLike]]:
    """chunk_files + run_mercat2 in one step with no chunk files: the file is read (inflated)
    once, the reference's cut points are computed over the bytes, and each byte range is
    counted as one chunk (filtered on its own).  Same TSV as the two-step path.

    ``streams`` contexts (HIP streams) count different chunks concurrently -- the host-to-device
    copy and parse of one chunk overlap the LDS-bound counting of another -- and are summed on
    the device at the end (mk_merge_from).  ``canonical`` is the opt-in extension of
    mk_set_canonical (not reference behaviour)."""
    from concurrent.futures import ThreadPoolExecutor
    data = map_fasta(file)  # plain files: memory-mapped, no host copy
    chunked = chunk_mib > 0 and os.stat(file).st_size >= chunk_mib * 1024 * 1024
    offs = chunk_offsets(data, chunk_mib * 1024 * 1024) if chunked else [0, len(data)]
    chunks = list(zip(offs[:-1], offs[1:]))
    view = memoryview(data)
    alphabet = guess_alphabet(file, data)
    n = max(1, min(int(streams), len(chunks)))
    ctxs = [native.Counter(kmer, alphabet, device, canonical=canonical and alphabet == native.ALPHABET_NT2) for _ in range(n)]
    try:
        def share(i):
            for a, b in chunks[i::n]:
                ctxs[i].count_chunk(view[a:b], min_count)
        if n == 1:
            share(0)
        else:
            with ThreadPoolExecutor(n) as pool:  # ctypes releases the GIL inside the ABI calls
                list(pool.map(share, range(n)))
            for other in ctxs[1:]:
                ctxs[0].merge_from(other)
        return _finish(ctxs[0], basename, out_file)
    finally:
        for c in ctxs:
            c.close()
