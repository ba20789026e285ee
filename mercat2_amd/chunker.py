"""Drop-in for MerCat2's ``mercat2_Chunker`` (lib/mercat2_Chunker.py).

The reference splits a FASTA into files of >= chunksize bytes at lines containing '>' and
hands the files to find_kmers.  Here the cut points are computed once over the decompressed
bytes (``chunk_offsets``: the "virtual chunker", native code in libmercat_hip.so) so the count
path can feed the same byte ranges to the GPU without writing them; ``Chunker`` still writes the
files, byte-identical to the reference's, for callers that want them on disk.
"""
from __future__ import annotations

import glob
import os
from pathlib import Path
from typing import List, Union

import numpy as np

from . import native
from .kmers import read_fasta_bytes


def human2bytes(s: str) -> int:
    """'100M' -> 104857600; same grammar as lib/mercat2_Chunker.py:82-139."""
    tables = (("B", "K", "M", "G", "T", "P", "E", "Z", "Y"),
              ("byte", "kilo", "mega", "giga", "tera", "peta", "exa", "zetta", "iotta"),
              ("Bi", "Ki", "Mi", "Gi", "Ti", "Pi", "Ei", "Zi", "Yi"),
              ("byte", "kibi", "mebi", "gibi", "tebi", "pebi", "exbi", "zebi", "yobi"))
    i = 0
    while i < len(s) and (s[i].isdigit() or s[i] == "."):
        i += 1
    number, unit = float(s[:i]), s[i:].strip()
    if unit == "k":
        unit = "K"
    for t in tables:
        if unit in t:
            return int(number * (1 << (10 * t.index(unit))))
    raise ValueError("can't interpret %r" % s)


def chunk_offsets(data, chunksize: int) -> List[int]:
    """[0, cut1, cut2, ..., len(data)]: chunk i is data[offs[i]:offs[i+1]]."""
    cuts = native.chunk_cuts(data, int(chunksize))
    return [0] + [int(c) for c in cuts] + [len(data)]


def _normalise_newlines(b: bytes) -> bytes:
    """What text-mode read + write does to a chunk: '\\r\\n' and lone '\\r' become '\\n'."""
    return b.replace(b"\r\n", b"\n").replace(b"\r", b"\n")


class Chunker:
    """Chunker(path, dest, chunksize='1000M', delim='>') -- same constructor and ``files``
    attribute as the reference class (only the delimiter mode the count path uses)."""

    def __init__(self, path, dest, chunksize: Union[str, int] = "1000M", delim: str = ">", lines=None):
        if delim != ">" or lines is not None:
            raise NotImplementedError("only delim='>' (the mode MerCat2's count path uses) is provided")
        self.path = str(path)
        self.dest = dest
        self.chunksize = human2bytes(chunksize) if isinstance(chunksize, str) else int(chunksize)
        self.delim = delim
        self.fn = os.path.basename(self.path)
        self.name = Path(path).stem.split(".")[0]
        self.ext = "".join(Path(path).suffixes[:-1])
        os.makedirs(dest, exist_ok=True)
        data = read_fasta_bytes(path) if self.path.endswith(".gz") else Path(path).read_bytes()
        offs = chunk_offsets(data, self.chunksize)
        for i in range(len(offs) - 1):
            with open(os.path.join(dest, "%s.%05d%s" % (self.name, i, self.ext)), "wb") as w:
                w.write(_normalise_newlines(data[offs[i]:offs[i + 1]]))
        self.files = glob.glob(os.path.join(dest, "*"))
