import os, sys, time, tempfile
sys.path.insert(0, ".")
from mercat2_amd import native, harness
d = tempfile.mkdtemp(dir="/tmp")
path = os.path.join(d, "S.fna")
data = native.synth_reads(10_000_000, 3, 2_000_000, 150, 4)
open(path, "wb").write(memoryview(data))
for mib in (100, 10, 1):
    for rep in range(2):
        st = {}
        t0 = time.perf_counter()
        harness.run_sample("S", path, os.path.join(d, "o.tsv"), 31, 2, mib, stats=st, report=lambda s: None)
        dt = time.perf_counter() - t0
    print("-s %d: %d chunks, %.3f s (%.2f ms per chunk), wait io %.3f gpu %.3f" % (mib, st["chunks"], dt, dt / st["chunks"] * 1e3, st["s_wait_io"], st["s_wait_gpu"]))
