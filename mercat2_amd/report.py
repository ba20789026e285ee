"""Drop-in for the table-merging step of lib/mercat2_report.py (``merge_tsv``, lines 98-156, and
``merge_tsv_T``, lines 160-194): the combined sample x k-mer table that feeds the reference's plots
and PCA, and its transpose that beta diversity reads.

``merge_counters`` builds it straight from the samples' tables on the GPU (no TSV re-read);
``merge_tsv`` keeps the reference's signature (a dict of TSV paths) by loading the files into engine
tables first.  Header ``<first column>\\t<sorted names>``, then one row per k-mer.

Which rows: the reference's streaming merge looks for the next k-mer only among the samples that advanced in
the current step and writes a sample's pending count under the k-mer at hand whenever its own key is not
greater (lib/mercat2_report.py:131-150) -- a key held only by samples that did not advance gets no row of its
own and its count lands in a later row.  ``merge_tsv`` (the reference's name) and the CLI write exactly those
rows (``as_reference=True``), so the file is the one MerCat2 writes; ``as_reference=False`` gives the true union
(every k-mer of any sample, 0 where a sample lacks it).  For tables that share nearly all their keys -- k = 5
over genomes, the reference's committed runs -- the two are the same.  ``merge_tsv_T`` has no such quirk.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import numpy as np

from . import native


def merge_counters(counters: Dict[str, "native.Counter"], out_file, first_column: str = "k-mer",
                   as_reference: bool = True) -> int:
    """Write the combined table of ``{sample name: Counter}``; returns the number of rows written."""
    names = sorted(counters.keys())
    if not names:
        raise ValueError("merge_counters: no samples")
    return native.write_merged_tsv([counters[n] for n in names], names, out_file, first_column, as_reference)


def merge_counters_T(counters: Dict[str, "native.Counter"], out_file) -> int:
    """The transposed table (samples as rows) of ``{sample name: Counter}``; returns the number of k-mer columns.
    Columns are in sorted k-mer order (the reference's order is that of a Python set: arbitrary)."""
    names = sorted(counters.keys())
    if not names:
        raise ValueError("merge_counters_T: no samples")
    return native.write_merged_tsv_T([counters[n] for n in names], names, out_file)


def _load_tsv(path) -> tuple:
    """(first header field, kmers (rows, k) uint8, counts uint64) of a count table."""
    with open(path, "rb") as fh:
        head = fh.readline().decode().split("\t")[0]
        body = fh.read()
    if not body.strip():
        return head, np.zeros((0, 0), np.uint8), np.zeros(0, np.uint64)
    lines = body.split(b"\n")
    if not lines[-1]:
        lines.pop()
    k = lines[0].index(b"\t")
    flat = np.frombuffer(b"".join(l[:k] for l in lines), dtype=np.uint8).reshape(len(lines), k)
    counts = np.array([int(l[k + 1:]) for l in lines], dtype=np.uint64)
    return head, flat, counts


def merge_tsv_T(tsv_list: Dict[str, os.PathLike], out_file: os.PathLike, *, device: int = 0) -> None:
    """merge_tsv_T(tsv_list, out_file) of lib/mercat2_report.py:160-194 (same arguments): ``sample`` + one column per
    k-mer of any sample, one row per sample (sorted names), 0 where a sample lacks the k-mer."""
    _merge_files(tsv_list, out_file, device, transposed=True)


def merge_tsv(tsv_list: Dict[str, os.PathLike], out_file: os.PathLike, *, device: int = 0) -> None:
    """merge_tsv(tsv_list, out_file) of lib/mercat2_report.py:98-156 (same arguments)."""
    _merge_files(tsv_list, out_file, device, transposed=False)


def _merge_files(tsv_list, out_file, device: int, transposed: bool) -> None:
    names = sorted(tsv_list.keys())
    header: Optional[str] = None
    ctxs = []
    try:
        loaded = []
        for name in names:
            head, kmers, counts = _load_tsv(tsv_list[name])
            if header is None:
                header = head
            loaded.append((kmers, counts))
        k = next((km.shape[1] for km, _ in loaded if km.size), 1)
        for kmers, counts in loaded:
            c = native.Counter(k, native.ALPHABET_RAW, device)
            ctxs.append(c)
            c.import_exotic(kmers, counts)
        if transposed:
            native.write_merged_tsv_T(ctxs, names, out_file)
        else:
            native.write_merged_tsv(ctxs, names, out_file, header or "k-mer", as_reference=True)
    finally:
        for c in ctxs:
            c.close()
