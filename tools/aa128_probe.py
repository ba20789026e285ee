"""Protein k-mers of 13..25 residues on a real proteome (tests/golden/inputs/Rleg_pro.faa.gz, 2.3 M residues, repeated to
make a 100 MB-class chunk): packed two-word keys (default) against text rows (MK_NO_AA128=1)."""
import gzip, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mercat2_amd import native
raw = gzip.open(Path(__file__).resolve().parents[1] / "tests/golden/inputs/Rleg_pro.faa.gz").read()
residues = sum(len(l) for l in raw.split(b"\n") if not l.startswith(b">"))
big = int(sys.argv[1]) if len(sys.argv) > 1 else 1
if big > 1:  # (the same proteome `big` times over, record names aside: duplicates, as a read set would have)
    raw = raw * big
    residues *= big
for k in (13, 20, 25):
    with native.Counter(k, native.ALPHABET_AA5) as ctx:
        ctx.count_chunk(raw[:20000], 1)
        best = 1e9
        for rep in range(3):
            ctx.reset()
            t0 = time.perf_counter()
            ctx.count_chunk(raw, 1)
            rows = ctx.rows()
            best = min(best, time.perf_counter() - t0)
        ctx.reset(); ctx.reset_stats(); ctx.set_profiling(True); ctx.count_chunk(raw, 1); st = ctx.stats()
        print("aa k=%d %s: %.2f ms  %.2f Gresidues/s  rows %d   kernels: parse %.2f pack %.2f count %.2f exotic %.2f filter %.2f ms" % (
            k, st["mode_name"], best * 1e3, residues / best / 1e9, rows, st["ms_parse"], st["ms_pack"], st["ms_count"], st["ms_exotic"], st["ms_filter"]), flush=True)
