"""Throughput probe of the non-headline modes (run on the GPU box)."""
import sys, time
sys.path.insert(0, ".")
import torch
from mercat2_amd import native

def run(label, data_t, n, k, alpha, c=10, reps=3):
    ctx = native.Counter(k, alpha)
    ctx.count_device(data_t.data_ptr(), n, c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.reset()
        ctx.count_device(data_t.data_ptr(), n, c)
    rows = ctx.rows()
    dt = (time.perf_counter() - t0) / reps
    st = ctx.stats()
    print("%-28s mode=%-7s %.1f ms  %.2f Gbases/s rows=%d" % (label, st["mode_name"], dt * 1e3, st["symbols"] / (reps + 1) / dt / 1e9, rows))
    ctx.close()

host = native.synth_reads(10_000_000, 3, 600_000, 150, 4)
t = torch.from_numpy(host).cuda()
n = host.nbytes
for k in (3, 7, 12, 15, 17, 21, 31, 32, 33, 63, 100):
    run("nt 600k reads k=%d" % k, t, n, k, native.ALPHABET_NT2)
import numpy as np
rng = np.random.default_rng(1)
aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
prot = b"".join(b">p%d\n" % i + aa[rng.integers(0, 20, 300)].tobytes() + b"*\n" for i in range(200_000))
tp = torch.from_numpy(np.frombuffer(prot, dtype=np.uint8).copy()).cuda()
for k in (3, 5, 12, 13):
    run("aa 200k proteins k=%d" % k, tp, len(prot), k, native.ALPHABET_AA5, c=1)
