"""Amplicon-like input: the same few reads hundreds of thousands of times (a handful of hot buckets)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from mercat2_amd import native
from oracle import c_oracle
rng = np.random.default_rng(3)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
variants = [acgt[rng.integers(0, 4, 150)].tobytes() for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
pick = rng.integers(0, len(variants), n)
data = b"".join(b">r%d\n" % i + variants[pick[i]] + b"\n" for i in range(n))
for k in (31, 21, 40):
    with native.Counter(k, native.ALPHABET_NT2) as ctx:
        ctx.count_chunk(data[:2000], 1); ctx.reset()
        t0 = time.perf_counter()
        ctx.count_chunk(data, 10)
        km, cn = ctx.export()
        dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    okm, ocn = c_oracle.count(data, k, 10)
    print("k=%d: %.1f MB, gpu %.3f s, rows %d, oracle %.2f s, equal %s" % (k, len(data) / 1e6, dt, km.shape[0], time.perf_counter() - t1,
          np.array_equal(km, okm) and np.array_equal(cn, ocn)), flush=True)
