"""Per-kernel-family HIP-event times of one context over S3-shaped chunks (100 MiB of 150-bp reads from a 50 Mbp genome,
k=63, -c 10): partition and count kernel of the two-word path; MK_NO_PREFILTER=1 / MK_FORCE_PREFILTER=1 for A/B runs.
python tools/k63_probe.py [chunks] [genome] [min_count]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mercat2_amd import native

chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 6
genome = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
c = int(sys.argv[3]) if len(sys.argv) > 3 else 10
bufs = [torch.from_numpy(native.synth_reads(genome, 6, 650_000, 150, 7, 0, i * 650_000)).cuda() for i in range(min(chunks, 4))]
with native.Counter(63, native.ALPHABET_NT2) as ctx:
    for rep in range(2):
        ctx.reset()
        ctx.set_profiling(rep == 1)
        for i in range(chunks):
            b = bufs[i % len(bufs)]
            ctx.count_device(b.data_ptr(), b.numel(), c)
    st = ctx.stats()
    print("k=63 genome %d c=%d: parse %.1f us  part %.1f us  count %.1f us  filter %.1f us  rows %d  windows/chunk %d distinct/chunk %d" % (
        genome, c, 1e3 * st["ms_parse"] / max(1, st["n_parse"]), 1e3 * st["ms_part"] / max(1, st["n_part"]),
        1e3 * st["ms_count"] / max(1, st["n_count"]), 1e3 * st["ms_filter"] / max(1, st["n_filter"]), st["rows"],
        st["windows"] // max(1, st["chunks"]), st["distinct"] // max(1, st["chunks"])))
