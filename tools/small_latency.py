"""Latency of one small sample through the product host layer (harness.run_sample), on the GPU box."""
import gzip, os, sys, tempfile, time
sys.path.insert(0, ".")
from mercat2_amd import harness, native
d = tempfile.mkdtemp(dir="/tmp")
src = os.path.join(d, "RW1_pro.faa")
open(src, "wb").write(gzip.open("tests/golden/inputs/RW1_pro.faa.gz", "rb").read())
nt = os.path.join(d, "A.fasta")
open(nt, "wb").write(open("tests/golden/inputs/A.fasta", "rb").read())
for path, k in ((src, 5), (src, 3), (nt, 31), (nt, 5)):
    ts = []
    for i in range(12):
        t0 = time.perf_counter()
        harness.run_sample("s", path, os.path.join(d, "o.tsv"), k, 2, report=lambda s: None)
        ts.append(time.perf_counter() - t0)
    print("%s k=%d: first %.1f ms, then median %.2f ms (min %.2f)" % (os.path.basename(path), k, ts[0] * 1e3, sorted(ts[1:])[len(ts) // 2] * 1e3, min(ts) * 1e3))
    # where it goes: context creation alone
    t0 = time.perf_counter()
    for i in range(10):
        native.Counter(k, native.ALPHABET_AA5 if path.endswith("faa") else native.ALPHABET_NT2).close()
    print("   create+destroy a context: %.2f ms" % ((time.perf_counter() - t0) * 100))
