import os, sys, time
sys.path.insert(0, ".")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
from mercat2_amd import native
from mercat2_amd.chunker import chunk_offsets
from mercat2_amd import dist as mkdist
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
k = 31
host = native.synth_reads(10_000_000, 3, 10_000_000, 150, 4)
offs = chunk_offsets(host, 100 << 20)
text = torch.from_numpy(host).to(dev)
ctx = native.Counter(k, native.ALPHABET_NT2, device=0)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    ctx.reset()
    for a, b in zip(offs[:-1], offs[1:]):
        ctx.count_device(text.data_ptr() + a, b - a, 10)
    t = [T()]
    words = ctx.words_per_key()
    cap = ctx.rows() + 1; t.append(T())
    keys = torch.empty((cap, words), dtype=torch.int64, device=dev); cnts = torch.empty(cap, dtype=torch.int64, device=dev); t.append(T())
    n = ctx.export_pairs_device(keys.data_ptr(), cnts.data_ptr(), cap); t.append(T())
    rows = torch.cat([keys[:n], cnts[:n, None]], dim=1); t.append(T())
    ex_k, ex_c = ctx.export_exotic(); t.append(T())
    ctx.reset(); t.append(T())
    got, extras = mkdist.exchange_rows(rows, 2 * k, int(ex_c.size), None, True); t.append(T())
    rk = got[:, :words].contiguous(); rc = got[:, words].contiguous(); t.append(T())
    ctx.import_pairs_device(rk.data_ptr(), rc.data_ptr(), got.shape[0]); t.append(T())
    r = ctx.rows(); t.append(T())
    names = ["rows()", "alloc", "export", "cat", "export_exotic", "reset", "exchange", "split copies", "import", "rows()"]
    print("rep", rep, " ".join("%s %.0fus" % (nm, (b - a) * 1e6) for nm, a, b in zip(names, t[:-1], t[1:])), "total %.2f ms" % ((t[-1] - t[0]) * 1e3), flush=True)
dist.destroy_process_group()
