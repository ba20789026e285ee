"""Odd input shapes through the engine, timed, against the C oracle (run on the GPU box)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from mercat2_amd import native
from oracle import c_oracle
rng = np.random.default_rng(1)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
rand = acgt[rng.integers(0, 4, n)].tobytes()
shapes = {
    "one_long_line": b">x\n" + rand + b"\n",
    "tiny_records": b"".join(b">r%d\n" % i + rand[i * 31:i * 31 + 31] + b"\n" for i in range(n // 80)),
    "all_N": b">n\n" + b"N" * (n // 2) + b"\n",
    "soft_masked": b">m\n" + b"".join((rand[i:i + 5000].lower() if (i // 5000) % 2 else rand[i:i + 5000]) for i in range(0, n // 2, 5000)) + b"\n",
    "wrapped_60": b">w\n" + b"\n".join(rand[i:i + 60] for i in range(0, n, 60)) + b"\n",
    "crlf_wrapped": b">w\r\n" + b"\r\n".join(rand[i:i + 70] for i in range(0, n // 2, 70)) + b"\r\n",
    "blank_in_lines": b">b\n" + b"\n".join(rand[i:i + 30] + b" " + rand[i + 30:i + 60] for i in range(0, n // 4, 60)) + b"\n",
}
for name, data in shapes.items():
    for k in (21, 31):
        with native.Counter(k, native.ALPHABET_NT2) as ctx:
            ctx.count_chunk(data[:1000], 1)   # warm the context
            ctx.reset()
            t0 = time.perf_counter()
            ctx.count_chunk(data, 2)
            dt = time.perf_counter() - t0
            km, cn = ctx.export()
        t1 = time.perf_counter()
        okm, ocn = c_oracle.count(data, k, 2)
        print("%-15s k=%d %6.1f MB: gpu %.3f s (%.2f GB/s), rows %d, oracle %.1f s, equal %s" % (
            name, k, len(data) / 1e6, dt, len(data) / dt / 1e9, km.shape[0], time.perf_counter() - t1,
            np.array_equal(km, okm) and np.array_equal(cn, ocn)), flush=True)
