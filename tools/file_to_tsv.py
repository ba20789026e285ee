"""File-to-TSV timing: the reference's `Time to count` window -- FASTA on local disk
(page cache) -> TSV closed -- through the product host layer (harness.run_sample). Run on the GPU box."""
import os, sys, time, tempfile
sys.path.insert(0, ".")
from mercat2_amd import native, harness

def main():
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
    d = tempfile.mkdtemp(dir="/tmp")
    path = os.path.join(d, "S2.fna")
    t0 = time.perf_counter()
    data = native.synth_reads(10_000_000, 3, reads, 150, 4)
    with open(path, "wb") as f:
        f.write(memoryview(data))
    print("generated %s (%.2f GB) in %.1f s" % (path, data.nbytes / 1e9, time.perf_counter() - t0))
    del data
    gz = None
    if "--gz" in sys.argv:
        import subprocess
        t0 = time.perf_counter()
        subprocess.run(["gzip", "-1", "-k", path], check=True)
        gz = path + ".gz"
        print("gzip -1 -> %.2f GB in %.1f s" % (os.path.getsize(gz) / 1e9, time.perf_counter() - t0))
    for f in [path] + ([gz] if gz else []):
        for streams, threads in (((1, 0), (2, 0), (2, 2), (2, 16)) if not f.endswith('.gz') else ((2, 0), (2, 0), (2, 0))):
            for rep in range(2):
                out = os.path.join(d, "S2_counts_%d.tsv" % streams)
                st = {}
                t0 = time.perf_counter()
                harness.run_sample("S2", f, out, k, 10, 100, streams=streams, threads=threads, stats=st)
                dt = time.perf_counter() - t0
                print("%s streams=%d threads=%d rep=%d file-to-TSV %.3f s = %.2f Gbases/s (tsv %.1f MB) read+count %.3f s, wait io %.3f gpu %.3f, chunks %d"
                      % (os.path.basename(f), streams, st["threads"], rep, dt, reads * 150 / dt / 1e9, os.path.getsize(out) / 1e6,
                         st["s_total"], st["s_wait_io"], st["s_wait_gpu"], st["chunks"]))

main()
