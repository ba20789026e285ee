"""Per-kernel-family HIP-event times of one context over S2-shaped chunks (100 MiB of 150-bp reads, k=31, -c 10):
the parser alone (ms_parse / n_parse), the partition, the count kernel.  For A/B runs of parser variants:
MERCAT_HIP_LIB=build/libmercat_<tag>.so python tools/parse_probe.py [chunks]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mercat2_amd import native

chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 6
text = native.synth_reads(10_000_000, 1, 660_000, 150, 2)
buf = torch.from_numpy(text).cuda()
import os
with native.Counter(31, native.ALPHABET_NT2, canonical=bool(os.environ.get('MK_PROBE_CANON'))) as ctx:
    for rep in range(2):
        ctx.reset()
        ctx.set_profiling(rep == 1)
        for i in range(chunks):
            try:
                ctx.count_device(buf.data_ptr(), buf.numel(), 10)
            except Exception as e:  # (ablation builds produce nonsense downstream of the parser)
                print("chunk failed:", str(e)[:80])
    st = ctx.stats()
    print("bytes/chunk %d  parse %.1f us  part %.1f us  count %.1f us  filter %.1f us  rows %d  reused %d  records/chunk %d  retries %d" % (
        buf.numel(), 1e3 * st["ms_parse"] / max(1, st["n_parse"]), 1e3 * st["ms_part"] / max(1, st["n_part"]),
        1e3 * st["ms_count"] / max(1, st["n_count"]), 1e3 * st["ms_filter"] / max(1, st["n_filter"]), st["rows"], st["part_reused"], st["records"] // max(1, st["chunks"]), st["part_retries"]))
