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
    for streams in (1, 2):
        for rep in range(2):
            out = os.path.join(d, "S2_counts_%d.tsv" % streams)
            t0 = time.perf_counter()
            harness.run_sample("S2", path, out, k, 10, 100, streams=streams)
            dt = time.perf_counter() - t0
            print("streams=%d rep=%d file-to-TSV %.2f s = %.2f Gbases/s (tsv %.1f MB)" % (streams, rep, dt, reads * 150 / dt / 1e9, os.path.getsize(out) / 1e6))

main()
