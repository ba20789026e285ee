"""Export / TSV time for large tables, one-word and two-word keys (run on the GPU box)."""
import os, sys, time, tempfile
sys.path.insert(0, ".")
from mercat2_amd import native
d = tempfile.mkdtemp(dir="/tmp")
data = native.synth_reads(5_000_000, 3, 1_000_000, 150, 4).tobytes()
for k in (31, 40, 63):
    with native.Counter(k, native.ALPHABET_NT2) as ctx:
        t0 = time.perf_counter(); ctx.count_chunk(data, 2); ctx.rows(); t1 = time.perf_counter()
        rows = ctx.write_tsv(os.path.join(d, "o.tsv"), "s"); t2 = time.perf_counter()
    print("k=%d: count %.3f s, write_tsv %.3f s for %d rows (%.0f MB)" % (k, t1 - t0, t2 - t1, rows, os.path.getsize(os.path.join(d, "o.tsv")) / 1e6), flush=True)
