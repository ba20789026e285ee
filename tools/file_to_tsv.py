"""File-to-TSV timing: the reference's `Time to count` window -- FASTA on local disk
(page cache) -> TSV closed -- through the product host layer (harness.run_sample). Run on the GPU box."""
import os, sys, time, tempfile
sys.path.insert(0, ".")
from mercat2_amd import native, harness

def _bgzf_member(p):
    import struct, zlib
    c = zlib.compressobj(1, zlib.DEFLATED, -15)
    body = c.compress(p) + c.flush()
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 12 + 6 + len(body) + 8 - 1) + body +
            struct.pack("<II", zlib.crc32(p), len(p)))


def _bgzf_span(span):
    path, a, b = span
    with open(path, "rb") as f:
        f.seek(a)
        data = f.read(b - a)
    return b"".join(_bgzf_member(data[i:i + 65280]) for i in range(0, len(data), 65280))


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
    bg = None
    if "--bgzf" in sys.argv:
        from multiprocessing import Pool
        t0 = time.perf_counter()
        size = os.path.getsize(path)
        spans = [(path, a, min(a + (64 << 20), size)) for a in range(0, size, 64 << 20)]
        with Pool(16) as pool:
            parts = pool.map(_bgzf_span, spans)
        bg = os.path.join(d, "S2_bgzf.fna.gz")
        with open(bg, "wb") as w:
            for part in parts:
                w.write(part)
            w.write(_bgzf_member(b""))
        print("bgzf -> %.2f GB in %.1f s" % (os.path.getsize(bg) / 1e9, time.perf_counter() - t0))
    for f in [path] + ([gz] if gz else []) + ([bg] if bg else []):
        for streams, threads in (((1, 0), (2, 0), (2, 2), (2, 16)) if not f.endswith('.gz') else (((2, 0), (2, 16), (2, 4), (2, 1)) if 'bgzf' not in f else ((2, 1), (2, 4), (2, 8), (2, 16)))):
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
