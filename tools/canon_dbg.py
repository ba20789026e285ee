import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from mercat2_amd import native
from mercat2_amd.chunker import chunk_offsets
data = native.synth_reads(10_000_000, 3, 10_000_000, 150, 4)
offs = chunk_offsets(data, 100 * 1024 * 1024)
view = memoryview(data)
for canon in (False, True):
    for c in (1, 10):
        with native.Counter(31, native.ALPHABET_NT2, canonical=canon) as ctx:
            for lo, hi in list(zip(offs[:-1], offs[1:]))[:3]:
                ctx.count_chunk(view[lo:hi], c)
            st = ctx.stats()
            print("canon", canon, "c", c, "rows", ctx.rows(), {k: st[k] for k in ("windows", "survivors", "rows", "part_retries", "part_reused", "records", "mode_name") if k in st}, flush=True)
