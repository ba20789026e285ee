"""Randomised differential run at read-set sizes (sampled histograms, several chunks, two contexts) against the
C oracle's per-chunk composition.  Run on the GPU box."""
import os, random, sys, tempfile
sys.path.insert(0, ".")
import numpy as np
from mercat2_amd import native
from mercat2_amd.chunker import chunk_offsets
from oracle import c_oracle, cpu_ref

CANON = "--canonical" in sys.argv  # the opt-in extension: oracle = the chunk's counts at c = 0 folded onto min(key, revcomp), then the filter
sys.argv = [a for a in sys.argv if a != "--canonical"]

rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
d = tempfile.mkdtemp(dir="/tmp")
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    genome = rng.choice([50_000, 300_000, 2_000_000, 5_000_000])
    reads = rng.choice([60_000, 200_000, 500_000, 900_000])
    sub = rng.choice([0, 0, 1000, 20000])
    k = rng.choice([18, 21, 25, 31, 32, 33, 40, 55, 64])
    c = rng.choice([1, 2, 5, 10])
    mib = rng.choice([0, 8, 24, 64])
    data = native.synth_reads(genome, rng.randrange(1 << 30), reads, 150, rng.randrange(1 << 30), sub).tobytes()
    if rng.random() < 0.3:   # an N run and a homopolymer somewhere
        at = rng.randrange(len(data) // 2)
        at = data.index(b"\n>", at) + 1
        data = data[:at] + b">gap\n" + b"ACGT" * 200 + b"N" * rng.choice([10, 5000]) + b"A" * rng.choice([40, 30000]) + b"\n" + data[at:]
    path = os.path.join(d, "r%d.fna" % case)
    open(path, "wb").write(data)
    size = mib << 20
    offs = chunk_offsets(data, size) if size and len(data) >= size else [0, len(data)]
    want = {}
    for a, b in zip(offs[:-1], offs[1:]):
        if CANON:
            table = {key: n for key, n in cpu_ref.canonical_fold(c_oracle.count_dict(data[a:b], k, 0)).items() if n >= c}
        else:
            table = c_oracle.count_dict(data[a:b], k, c)
        for key, n in table.items():
            want[key] = want.get(key, 0) + n
    ctxs = [native.Counter(k, native.ALPHABET_NT2, canonical=CANON) for _ in range(rng.choice([1, 2, 3]))]
    shared = False
    if len(ctxs) > 1 and k <= 32 and rng.random() < 0.6:  # one running table for the contexts (mk_share_table)
        for x in ctxs[1:]:
            x.share_table(ctxs[0])
        shared = True
    try:
        if rng.random() < 0.5:  # a first sample through the same contexts: hints and table sizes from another text
            other = native.synth_reads(rng.choice([40_000, 3_000_000]), rng.randrange(1 << 30), 100_000, 150, rng.randrange(1 << 30)).tobytes()
            p2 = os.path.join(d, "w%d.fna" % case)
            open(p2, "wb").write(other)
            native.count_file(ctxs, p2, 4 << 20, rng.choice([1, 2, 10]))
            for x in ctxs:
                x.reset()
        st = native.count_file(ctxs, path, size, c)
        got = ctxs[0].to_dict()
        retries = sum(x.stats()["part_retries"] for x in ctxs)
    finally:
        for x in ctxs:
            x.close()
    ok = got == want
    print("case %d: genome %d reads %d sub %d k %d c %d -s %d ctxs %d%s: %d chunks, %d rows, retries %d, %s" % (
        case, genome, reads, sub, k, c, mib, len(ctxs), " shared" if shared else "", st["chunks"], len(got), retries, "ok" if ok else "MISMATCH"), flush=True)
    if not ok:
        sys.exit(1)
print("all ok")
