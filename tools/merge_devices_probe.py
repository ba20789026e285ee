import os, sys, time
sys.path.insert(0, ".")
import torch
from mercat2_amd import native
from mercat2_amd.chunker import chunk_offsets
k = 31
host = native.synth_reads(10_000_000, 3, 10_000_000, 150, 4)
offs = chunk_offsets(host, 100 << 20)
text = torch.from_numpy(host).cuda()
spans = list(zip(offs[:-1], offs[1:]))
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for n_ctx in (2, 4):
    ctxs = [native.Counter(k, native.ALPHABET_NT2, device=0) for _ in range(n_ctx)]
    for rep in range(3):
        for i, c in enumerate(ctxs):
            c.reset()
            for a, b in spans[i::n_ctx]:
                c.count_device(text.data_ptr() + a, b - a, 10)
        t0 = T()
        st = native.merge_devices(ctxs, native.MERGE_RANGES | (native.MERGE_BALANCED if rep else 0))
        t1 = T()
        print("contexts=%d rep %d: %.2f ms" % (n_ctx, rep, (t1 - t0) * 1e3), flush=True)
