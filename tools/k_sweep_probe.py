"""Every k regime on the same reads, timed (run on the GPU box)."""
import sys, time
sys.path.insert(0, ".")
from mercat2_amd import native
data = native.synth_reads(1_000_000, 1, 600_000, 150, 2).tobytes()
prot = bytes((b"ACDEFGHIKLMNPQRSTVWY" * 7)[(i * 7919) % 140] for i in range(3_000_000))
prot = b">p\n" + b"\n".join(prot[i:i + 60] for i in range(0, len(prot), 60)) + b"\n"
for alpha, name, text, ks in ((native.ALPHABET_NT2, "nt", data, (1, 2, 7, 8, 11, 12, 17, 18, 25, 32, 33, 48, 64, 65, 100, 149, 150, 151)),
                              (native.ALPHABET_AA5, "aa", prot, (1, 3, 4, 5, 8, 12, 13, 20))):
    for k in ks:
        with native.Counter(k, alpha) as ctx:
            ctx.count_chunk(text[:2000], 1)
            ctx.reset()
            t0 = time.perf_counter()
            ctx.count_chunk(text, 2)
            rows = ctx.rows()
            dt = time.perf_counter() - t0
            mode = ctx.stats()["mode_name"]
        print("%s k=%-3d %-7s %.4f s  %6.2f Gbases/s  rows %d" % (name, k, mode, dt, (90e6 if name == "nt" else 3e6) / dt / 1e9, rows), flush=True)
