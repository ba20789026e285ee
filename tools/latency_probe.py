import sys, time
sys.path.insert(0, "/root/repo")
from pathlib import Path
from mercat2_amd import native, kmers, harness
p = Path("/root/repo/tests/golden/inputs")
for name, k in [("RW1.fna.gz", 31), ("RW1_pro.faa.gz", 3), ("A.fasta", 31), ("Scaffolds_with-NNN.fna.gz", 21)]:
    for rep in range(3):
        t0 = time.perf_counter()
        d = kmers.find_kmers(p / name, k, 10)
        t1 = time.perf_counter()
        harness.run_sample("x", p / name, "/tmp/x.tsv", k, 10, 100)
        t2 = time.perf_counter()
    print("%-28s k=%-3d find_kmers %.1f ms (%d keys)   run_sample %.1f ms" % (name, k, (t1 - t0) * 1e3, len(d), (t2 - t1) * 1e3))
t0 = time.perf_counter(); c = native.Counter(31); t1 = time.perf_counter(); c.close(); print("ctx create %.1f ms" % ((t1 - t0) * 1e3))
