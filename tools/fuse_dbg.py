"""Debug: chunks of the ingest test's text one by one through one context; after every chunk compare with the oracle's running sum."""
import io, random, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from mercat2_amd import native
from oracle import cpu_ref
from test_gpu_ingest import _reads
rng = random.Random(11)
text = _reads(rng, 4000)
k, c, chunk_bytes = 21, 2, 100_000
fh = io.TextIOWrapper(io.BytesIO(text), encoding="utf-8", newline=None)
groups = cpu_ref.split_lines(fh, chunk_bytes)
want = {}
with native.Counter(k, native.ALPHABET_NT2) as ctx:
    for i, g in enumerate(groups):
        part = cpu_ref.count_lines(g, k, c)
        for key, n in part.items():
            want[key] = want.get(key, 0) + n
        data = "".join(g).encode()
        ctx.count_chunk(data, c)
        got = ctx.to_dict()
        st = ctx.stats()
        bad = [(key, got.get(key), n) for key, n in want.items() if got.get(key) != n]
        extra = [key for key in got if key not in want]
        print("chunk", i, "bytes", len(data), "survivors(oracle)", len(part), "rows", len(got), "want", len(want), "wrong", len(bad), "extra", len(extra),
              "fused", st["fused_chunks"], "spilled", st["fuse_spilled"], bad[:3], flush=True)
