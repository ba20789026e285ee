"""GPU box: protein k-mers through the 8-byte-key partition (4 <= k <= 12) on 200 k random proteins of 300 residues --
for rocprofv3 --kernel-trace --stats (which kernel of that path the time goes to).  python tools/aa_probe.py [k] [c]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from mercat2_amd import native
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
c = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(1)
aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
prot = b"".join(b">p%d\n" % i + aa[rng.integers(0, 20, 300)].tobytes() + b"*\n" for i in range(200_000))
tp = torch.from_numpy(np.frombuffer(prot, dtype=np.uint8).copy()).cuda()
with native.Counter(k, native.ALPHABET_AA5) as ctx:
    ctx.count_device(tp.data_ptr(), len(prot), c)
    torch.cuda.synchronize()
    ctx.reset_stats()
    ctx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.reset()
        ctx.count_device(tp.data_ptr(), len(prot), c)
    rows = ctx.rows()
    dt = (time.perf_counter() - t0) / 3
    st = ctx.stats()
print("aa k=%d c=%d: %.2f ms per chunk, %.1f Gresidues/s, rows %d, mode %s; HIP events per chunk: parse %.0f pack %.0f part %.0f count %.0f filter %.0f us"
      % (k, c, dt * 1e3, st["symbols"] / 3 / dt / 1e9, rows, st["mode_name"], *(1e3 * st["ms_" + n] / 3 for n in ("parse", "pack", "part", "count", "filter"))))
