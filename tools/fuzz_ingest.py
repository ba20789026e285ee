"""Randomised differential run: mk_count_file (random k, -c, chunk size, reader block size, plain/gzip/BGZF-less,
1-3 contexts, texts with every quirk) against the oracle's composition.  Run on the GPU box."""
import gzip, io, os, random, sys, tempfile
sys.path.insert(0, ".")
import numpy as np
from mercat2_amd import native
from oracle import cpu_ref

def text(rng):
    alpha = rng.choice([b"ACGT", b"ACGT", b"ACGTN", b"ACGTacgtN", b"ACDEFGHIKLMNPQRSTVWY", b"ACGT*"])
    out = []
    for i in range(rng.randint(0, 60)):
        n = rng.choice([0, 1, 5, 30, 31, 32, 33, 64, 65, 150, 400, 2000])
        seq = bytes(rng.choice(alpha) for _ in range(n))
        w = rng.choice([0, 0, 60, 70, 1, 7])
        lines = [seq[j:j + w] for j in range(0, len(seq), w)] if w else [seq]
        nl = rng.choice([b"\n", b"\n", b"\n", b"\r\n", b"\r"])
        hdr = rng.choice([b">r%d" % i, b">r%d desc > more" % i, b"  >r%d" % i])
        body = [(b" " + l if rng.random() < 0.05 else l) for l in lines]
        out.append(hdr + nl + nl.join(body) + nl)
        if rng.random() < 0.1:
            out.append(nl)
    t = b"".join(out)
    if rng.random() < 0.3:
        t = t.rstrip(b"\r\n")
    if rng.random() < 0.15:
        t = b"ACGTACGTTGCA" + b"\n" + t
    return t

def oracle(data, k, c, chunk_bytes, disk):
    if chunk_bytes > 0 and disk >= chunk_bytes:
        fh = io.TextIOWrapper(io.BytesIO(data), encoding="utf-8", newline=None)
        return cpu_ref.merge_counts(cpu_ref.count_lines(g, k, c) for g in cpu_ref.split_lines(fh, chunk_bytes))
    return cpu_ref.count_text(data, k, c)

rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
d = tempfile.mkdtemp(dir="/tmp")
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 300
for case in range(cases):
    data = text(rng)
    gz = rng.random() < 0.4
    path = os.path.join(d, "f%d.%s" % (case, "fna.gz" if gz else "fna"))
    open(path, "wb").write(gzip.compress(data, rng.choice([1, 6, 9])) if gz else data)
    disk = os.path.getsize(path)
    k = rng.choice([1, 2, 3, 5, 7, 8, 11, 12, 15, 17, 18, 21, 25, 31, 32, 33, 40, 63, 64, 65, 70])
    c = rng.choice([1, 1, 2, 3, 10])
    chunk = rng.choice([0, 0, max(1, disk // 3), max(1, disk // 7), 10 * disk + 1, 50])
    os.environ["MK_INGEST_BLOCK"] = str(rng.choice([64, 1000, 4097, 1 << 16, 1 << 22]))
    alpha = rng.choice([native.ALPHABET_NT2, native.ALPHABET_NT2, native.ALPHABET_AA5, native.ALPHABET_RAW])
    nctx = rng.choice([1, 2, 3])
    want = oracle(data, k, c, chunk, disk)
    ctxs = [native.Counter(k, alpha) for _ in range(nctx)]
    try:
        native.count_file(ctxs, path, chunk, c, threads=rng.choice([0, 1, 3]))
        got = ctxs[0].to_dict()
    finally:
        for x in ctxs:
            x.close()
    if got != want:
        print("MISMATCH case", case, dict(k=k, c=c, chunk=chunk, gz=gz, alpha=alpha, nctx=nctx, block=os.environ["MK_INGEST_BLOCK"], n=len(data)))
        open("gpurun_out/fuzz_fail_%d.bin" % case, "wb").write(data)
        sys.exit(1)
print("ok:", cases, "cases")
