"""CPU oracle for the MerCat2 k-mer counting hot path.  TEST INFRASTRUCTURE ONLY.

This module is a from-scratch restatement, in plain Python, of the algorithm the
reference implements for its counting path.  It exists to *check* the HIP path; it is
never imported by the product package ``mercat2_amd`` (only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may use it).

Pinned: every function below is checked (tests/test_oracle_golden.py) against golden
vectors produced by importing the reference itself in the build container
(tests/golden/make_golden.py) and against the reference's own committed result tables
(results/2023-11-29/*/tsv_*/*_counts.tsv) -- see tests/golden/README.md.

Reference semantics restated (paths relative to the reference checkout):

* ``find_kmers``        lib/mercat2_kmers.py:32-78
* record loop           lib/mercat2_kmers.py:49-69  (strip, '>' header, '*' removal, de-wrap)
* min-count filter      lib/mercat2_kmers.py:73-76  (per file / per chunk, *before* any merge)
* ``Chunker``           lib/mercat2_Chunker.py:14-59 (stream_delim) and human2bytes :82-139
* ``chunk_files``       bin/mercat2.py:86-106
* ``run_mercat2``       bin/mercat2.py:115-137 (sum survivors, sorted TSV, no file when empty)
"""
from __future__ import annotations

import gzip
import io
import os
from collections import Counter
from pathlib import Path
from typing import Dict, Iterable, Iterator, List, Optional, Tuple, Union

PathLike = Union[str, os.PathLike]


# --------------------------------------------------------------------------- records
def _open_text(path: PathLike):
    """Text-mode open exactly as the reference does it: gzip iff the *last* suffix is
    '.gz' (lib/mercat2_kmers.py:47), default encoding, universal newlines."""
    p = Path(path)
    if p.suffix == ".gz":
        return gzip.open(p, "rt")
    return open(p, "r")


def iter_records(lines: Iterable[str]) -> Iterator[str]:
    """Yield the de-wrapped sequence of every record (lib/mercat2_kmers.py:49-69).

    A line is a header when, after ``str.strip()``, it starts with '>'.  Every other
    line is appended (stripped, with every '*' deleted) to the running record.  Text in
    front of the first header is a record of its own.  Empty records are not yielded
    (they contribute no window either way).
    """
    parts: List[str] = []
    for raw in lines:
        s = raw.strip()
        if s[:1] == ">":
            if parts:
                rec = "".join(parts)
                parts = []
                if rec:
                    yield rec
        else:
            parts.append(s.replace("*", ""))
    if parts:
        rec = "".join(parts)
        if rec:
            yield rec


def count_records(records: Iterable[str], k: int) -> Counter:
    """All length-k windows of every record, as substring keys (kmers.py:56-60,65-69)."""
    tally: Counter = Counter()
    for rec in records:
        n = len(rec) - k + 1
        if n > 0:
            tally.update(rec[i:i + k] for i in range(n))
    return tally


def apply_min_count(tally: Dict[str, int], min_count: int) -> Dict[str, int]:
    """Keep keys whose count is >= min_count (kmers.py:73-76)."""
    return {key: n for key, n in tally.items() if n >= min_count}


def count_lines(lines: Iterable[str], k: int, min_count: int) -> Dict[str, int]:
    return apply_min_count(count_records(iter_records(lines), k), min_count)


def count_text(data: Union[bytes, str], k: int, min_count: int) -> Dict[str, int]:
    """Count a FASTA held in memory.  ``bytes`` are decoded the way a text-mode file
    would be (utf-8, universal newlines), so '\\r\\n' and lone '\\r' end lines."""
    if isinstance(data, bytes):
        fh = io.TextIOWrapper(io.BytesIO(data), encoding="utf-8", newline=None)
    else:
        fh = io.StringIO(data, newline=None)
    return count_lines(fh, k, min_count)


def find_kmers(file: PathLike, kmer: int, min_count: int) -> Dict[str, int]:
    """Same contract as the reference's find_kmers (kmers.py:32-78)."""
    with _open_text(file) as fh:
        return count_lines(fh, kmer, min_count)


# --------------------------------------------------------------------------- chunker
_UNIT_TABLES = (
    ("B", "K", "M", "G", "T", "P", "E", "Z", "Y"),
    ("byte", "kilo", "mega", "giga", "tera", "peta", "exa", "zetta", "iotta"),
    ("Bi", "Ki", "Mi", "Gi", "Ti", "Pi", "Ei", "Zi", "Yi"),
    ("byte", "kibi", "mebi", "gibi", "tebi", "pebi", "exbi", "zebi", "yobi"),
)


def human2bytes(text: str) -> int:
    """'100M' -> 104857600 (lib/mercat2_Chunker.py:82-139): leading digits/dots are the
    number, the stripped remainder must be a unit symbol of one of the four tables ('k'
    is accepted for 'K'); unit i is 2**(10*i)."""
    i = 0
    while i < len(text) and (text[i].isdigit() or text[i] == "."):
        i += 1
    number = float(text[:i])
    unit = text[i:].strip()
    if unit == "k":
        unit = "K"
    for table in _UNIT_TABLES:
        if unit in table:
            return int(number * (1 << (10 * table.index(unit))))
    raise ValueError("can't interpret %r" % text)


def chunk_names(path: PathLike, n: int) -> List[str]:
    """File names the reference gives its chunks (Chunker.py:25-26,41): stem up to the
    first '.', a 5-digit index, then every suffix except the last."""
    p = Path(path)
    stem = p.stem.split(".")[0]
    ext = "".join(p.suffixes[:-1])
    return ["%s.%05d%s" % (stem, i, ext) for i in range(n)]


def split_lines(lines: Iterable[str], chunksize: int, delim: str = ">") -> List[List[str]]:
    """Group lines into chunks (Chunker.py:39-59): a line that *contains* ``delim``
    opens a new chunk when the bytes already written to the current chunk are
    >= chunksize.  Lines are the text-mode lines (newlines already normalised to '\\n'),
    sizes are their encoded lengths."""
    chunks: List[List[str]] = [[]]
    written = 0
    for line in lines:
        if delim in line and written >= chunksize:
            chunks.append([])
            written = 0
        chunks[-1].append(line)
        written += len(line.encode())
    return chunks


def chunk_file(path: PathLike, dest: PathLike, chunksize: Union[int, str], delim: str = ">") -> List[str]:
    """Write the chunks to ``dest`` like Chunker(...).files (sorted here; the reference
    returns them in glob order, which the count does not depend on)."""
    size = human2bytes(chunksize) if isinstance(chunksize, str) else int(chunksize)
    os.makedirs(dest, exist_ok=True)
    p = str(path)
    fh = gzip.open(p, "rt") if p.endswith(".gz") else open(p, "r")
    with fh:
        groups = split_lines(fh, size, delim)
    out = []
    for name, group in zip(chunk_names(path, len(groups)), groups):
        target = os.path.join(str(dest), name)
        with open(target, "w") as w:
            w.writelines(group)
        out.append(target)
    return out


def chunk_files(name: str, filename: PathLike, chunk_size: int, outpath: PathLike) -> Tuple[str, List[str]]:
    """bin/mercat2.py:86-106 -- chunk iff the on-disk size is >= chunk_size MiB."""
    if os.stat(filename).st_size >= chunk_size * 1024 * 1024:
        return name, chunk_file(filename, outpath, str(chunk_size) + "M", ">")
    return name, [str(filename)]


# --------------------------------------------------------------------------- harness
def merge_counts(tables: Iterable[Dict[str, int]]) -> Dict[str, int]:
    total: Counter = Counter()
    for t in tables:
        total.update(t)
    return dict(total)


def tsv_text(basename: str, table: Dict[str, int]) -> str:
    """Header 'k-mer\\t{basename}_Count' then rows in sorted(str) order (mercat2.py:130-133)."""
    rows = ["k-mer\t%s_Count\n" % basename]
    rows.extend("%s\t%d\n" % (key, table[key]) for key in sorted(table))
    return "".join(rows)


def run_mercat2(basename: str, files: List[PathLike], out_file: PathLike, kmer: int,
                min_count: int, num_cores: int = 1) -> Tuple[str, Optional[PathLike]]:
    """bin/mercat2.py:115-137: count every file with its own min_count filter, sum the
    survivors, write the sorted TSV; no file and a None path when nothing survives."""
    table = merge_counts(find_kmers(f, kmer, min_count) for f in files)
    if not table:
        print("No significant k-mers found")
        return basename, None
    print(f"Significant k-mers: {len(table)}")
    with open(out_file, "w") as w:
        w.write(tsv_text(basename, table))
    return basename, out_file


def count_sample_text(data: bytes, k: int, min_count: int, chunk_mib: int) -> Dict[str, int]:
    """The composition the CLI performs on one un-gzipped sample held in memory:
    chunk iff len(data) >= chunk_mib MiB (chunk_mib 0 = never), count every chunk with
    its own filter, sum."""
    if chunk_mib > 0 and len(data) >= chunk_mib * 1024 * 1024:
        fh = io.TextIOWrapper(io.BytesIO(data), encoding="utf-8", newline=None)
        groups = split_lines(fh, chunk_mib * 1024 * 1024)
        return merge_counts(count_lines(g, k, min_count) for g in groups)
    return count_text(data, k, min_count)


def merge_tsv_text(tables: Dict[str, Dict[str, int]], first: str = "k-mer") -> str:
    """The text merge_tsv writes for ``{sample name: table}`` (lib/mercat2_report.py:98-156), restated with its
    streaming loop as it is: the next k-mer is looked for only among the samples that advanced in the current step
    (:131, :149-150), and a sample whose pending key is not greater than the k-mer at hand has its count written
    under that k-mer, whatever its own key is (:137-140).  Pinned by tests/golden/report/*_merged.tsv."""
    names = sorted(tables)
    rows = {n: sorted(tables[n].items()) for n in names}
    pos = {n: 0 for n in names}
    out = [first + "\t" + "\t".join(names) + "\n"]
    kmer = sorted(rows[n][0][0] for n in names)[0]  # (an empty table makes the reference raise IndexError, too)
    while True:
        line = [kmer]
        nxt = set()
        for n in names:
            if pos[n] >= len(rows[n]) or rows[n][pos[n]][0] > kmer:
                line.append("0")
            else:
                line.append(str(rows[n][pos[n]][1]))
                pos[n] += 1
                if pos[n] < len(rows[n]):
                    nxt.add(rows[n][pos[n]][0])
        out.append("\t".join(line) + "\n")
        if not nxt:
            break
        kmer = sorted(nxt)[0]
    return "".join(out)


def union_tsv_text(tables: Dict[str, Dict[str, int]], first: str = "k-mer") -> str:
    """The true union table: every k-mer of any sample in sorted order, 0 where a sample lacks it."""
    names = sorted(tables)
    keys = sorted(set().union(*[set(t) for t in tables.values()])) if tables else []
    return first + "\t" + "\t".join(names) + "\n" + "".join(
        key + "\t" + "\t".join(str(tables[n].get(key, 0)) for n in names) + "\n" for key in keys)


def canonical_fold(table: Dict[str, int]) -> Dict[str, int]:
    """Opt-in extension (not reference behaviour): fold each ACGT key onto
    min(key, reverse-complement).  Keys with other letters are kept as they are."""
    comp = str.maketrans("ACGT", "TGCA")
    out: Counter = Counter()
    for key, n in table.items():
        if set(key) <= set("ACGT"):
            rc = key.translate(comp)[::-1]
            out[min(key, rc)] += n
        else:
            out[key] += n
    return dict(out)
