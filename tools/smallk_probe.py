"""Nucleotide k = 12 .. 20 on one S2-shaped chunk (100 MiB of 150-bp reads), -c 10, one context: Gbases/s of the whole
chunk pipeline.  MK_SK_MIN_K=<k> picks the smallest k that takes the super-k-mer partition (below: 8-byte-key partition)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mercat2_amd import native
text = native.synth_reads(10_000_000, 1, 660_000, 150, 2)
buf = torch.from_numpy(text).cuda()
for k in (12, 13, 14, 15, 16, 17, 18, 20):
    with native.Counter(k, native.ALPHABET_NT2) as ctx:
        for rep in range(2):
            ctx.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(4):
                ctx.count_device(buf.data_ptr(), buf.numel(), 10)
            rows = ctx.rows()
            dt = (time.perf_counter() - t0) / 4
        st = ctx.stats()
        print("k=%d %s  %.0f us per chunk  %.1f Gbases/s  rows %d records/chunk %d" % (k, st["mode_name"], dt * 1e6, 660_000 * 150 / dt / 1e9, rows, st["records"] // max(1, st["chunks"])), flush=True)
