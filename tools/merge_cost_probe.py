"""Cost of dist.merge_ranks at S2 size on RCCL with one rank (everything but the wire): export on the device, split
points, the two all_to_all_single calls (to itself), import, and the export that follows in bench.py's step."""
import os, sys, time
sys.path.insert(0, ".")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
from mercat2_amd import native
from mercat2_amd.chunker import chunk_offsets
from mercat2_amd import dist as mkdist
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 31
host = native.synth_reads(10_000_000, 3, 10_000_000, 150, 4)
offs = chunk_offsets(host, 100 << 20)
text = torch.from_numpy(host).to(dev)
ctx = native.Counter(k, native.ALPHABET_NT2, device=0)
cap = 21_000_000
ok = torch.empty(cap * ctx.words_per_key(), dtype=torch.int64, device=dev); oc = torch.empty(cap, dtype=torch.int64, device=dev)
for rep in range(3):
    ctx.reset()
    for a, b in zip(offs[:-1], offs[1:]):
        ctx.count_device(text.data_ptr() + a, b - a, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows = mkdist.merge_ranks(ctx, 2 * k, device=dev, always=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    n = ctx.export_pairs_device(ok.data_ptr(), oc.data_ptr(), cap)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("k=%d rep %d: rows %d merge_ranks %.2f ms, export after it %.2f ms" % (k, rep, rows, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
dist.destroy_process_group()
