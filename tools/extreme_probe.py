"""Low-complexity extremes (one k-mer millions of times) through the engine, timed, against the C oracle."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from mercat2_amd import native
from oracle import c_oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
data = b">a\n" + b"A" * n + b"\n>r\n" + b"ACGT" * (n // 4) + b"\n>t\n" + b"T" * (n // 10) + b"\n"
for k, c in ((31, 1), (32, 5), (21, 2), (40, 1)):
    t0 = time.time()
    with native.Counter(k, native.ALPHABET_NT2) as ctx:
        ctx.count_chunk(data, c)
        t1 = time.time()
        km, cn = ctx.export()
        st = ctx.stats()
    t2 = time.time()
    print("k=%d c=%d gpu count %.3f s, export %.3f s, rows %d, retries %d" % (k, c, t1 - t0, t2 - t1, km.shape[0], st["part_retries"]), flush=True)
    okm, ocn = c_oracle.count(data, k, c)
    print("   oracle %.2f s, equal %s" % (time.time() - t2, np.array_equal(km, okm) and np.array_equal(cn, ocn)), flush=True)
