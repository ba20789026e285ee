"""What the cross-GPU merge adds to a step at S2 size, everything but the wire (GPU box, one RCCL rank sending to itself):
(a) dist.merge_ranks as it is (bucket rows by owner on the device, two all_to_all_single calls, import of interleaved rows),
    phase by phase, equal and sampled owner bounds;
(b) the one-process path: native.merge_devices over contexts on the one device (bucket, device-to-device copies, import).
Usage: python tools/merge_phases_probe.py [world]   (world = how many owners the rows are bucketed for; default 8)"""
import os, sys, time
sys.path.insert(0, ".")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
from mercat2_amd import native
from mercat2_amd.chunker import chunk_offsets
from mercat2_amd import dist as mkdist
owners = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
k = 31
host = native.synth_reads(10_000_000, 3, 10_000_000, 150, 4)
offs = chunk_offsets(host, 100 << 20)
text = torch.from_numpy(host).to(dev)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
def count_into(ctx, spans):
    ctx.reset()
    for a, b in spans:
        ctx.count_device(text.data_ptr() + a, b - a, 10)
spans = list(zip(offs[:-1], offs[1:]))
for canon in (False, True):
    ctx = native.Counter(k, native.ALPHABET_NT2, device=0, canonical=canon)
    for rep in range(3):
        count_into(ctx, spans)
        t0 = T()
        rows = mkdist.merge_ranks(ctx, 2 * k, device=dev, always=True, balanced=canon)
        t1 = T()
        print("merge_ranks canonical=%d rep %d: rows %d  %.2f ms (one rank, RCCL to itself, %s bounds)" % (canon, rep, rows, (t1 - t0) * 1e3, "sampled" if canon else "equal"), flush=True)
    # the phases, by hand, bucketed for `owners` owners (what a rank of an N-GPU job does before the wire)
    for rep in range(2):
        count_into(ctx, spans)
        t = [T()]
        words = ctx.words_per_key()
        bounds = native.owner_bounds(2 * k, owners); t.append(T())
        cap = ctx.rows() + 1
        rows_t = torch.empty((cap, words + 1), dtype=torch.int64, device=dev); t.append(T())
        send = ctx.bucket_rows_device(bounds, rows_t.data_ptr(), cap); t.append(T())
        ex = ctx.export_exotic(); t.append(T())
        ctx.reset(); t.append(T())
        n = sum(send)
        out = torch.empty((n, words + 1), dtype=torch.int64, device=dev)
        dist.all_to_all_single(out, rows_t[:n], [n], [n]); t.append(T())
        ctx.import_rows_device(out.data_ptr(), n); t.append(T())
        names = ["bounds", "alloc", "bucket rows (hist + scatter)", "export_exotic", "reset", "all_to_all (self)", "import rows"]
        print("  phases canonical=%d rep %d (%d rows, %d owners, max share %.3f): " % (canon, rep, n, owners, max(send) / max(1, n)) +
              ", ".join("%s %.0f us" % (nm, (b - a) * 1e6) for nm, a, b in zip(names, t[:-1], t[1:])) + "; total %.2f ms" % ((t[-1] - t[0]) * 1e3), flush=True)
    ctx.close()
    # one process, several contexts on the one device: mk_merge_devices
    for n_ctx in (2, 8):
        ctxs = [native.Counter(k, native.ALPHABET_NT2, device=0, canonical=canon) for _ in range(n_ctx)]
        for rep in range(2):
            for i, c in enumerate(ctxs):
                count_into(c, spans[i::n_ctx])
            t0 = T()
            st = native.merge_devices(ctxs, native.MERGE_RANGES | native.MERGE_BALANCED)
            t1 = T()
            print("merge_devices canonical=%d contexts=%d rep %d: %.2f ms  rows_in %d rows_out %d moved %d max_owned %.3f  bucket %.2f copy+import %.2f ms" % (
                canon, n_ctx, rep, (t1 - t0) * 1e3, st["rows_in"], st["rows_out"], st["rows_moved"], st["max_owned"] / max(1, st["rows_out"]),
                st["s_bucket"] * 1e3, st["s_copy"] * 1e3), flush=True)
        for c in ctxs:
            c.close()
dist.destroy_process_group()
