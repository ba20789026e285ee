#!/usr/bin/env python3
"""Pin the HEADLINE workload's table to the reference: tests/golden/expected_s2.json.

Run only in the build container (needs /root/reference and ~40 GB of RAM, 15-30 minutes on 8 cores):

    python tests/golden/make_s2_golden.py [--procs 5] [--only s2|s3|s1|s3full]

What it does (bin/mercat2.py:86-106,115-137 composed by hand, since bin/mercat2.py needs Ray):
  S2  10,000,000 reads x 150 bp from a 10 Mbp genome (seeds 3/4: bench.py's own generator, mk_synth_reads)
      -> written as S2.fna -> the REFERENCE Chunker(path, dest, "100M", ">") cuts it into chunk files
      -> the REFERENCE find_kmers(chunk, 31, 1) per chunk (a process per chunk)
      -> forward table  = sum over chunks of {key: n | n >= 10}           (the -c 10 -s 100 run bench.py times)
      -> canonical table = sum over chunks of {fold(key): n | folded n >= 10}, fold = min(key, revcomp(key))
         (the opt-in mode's oracle is the reference's counts folded per chunk)
      find_kmers(chunk, 31, 10) is called directly on the last chunk and compared with the filter above.
  S3  the first two Chunker chunks of the 50 M-read sample (G = 50 Mbp, seeds 6/7), k = 63, -c 2:
      the two-word table at tens of millions of rows.
  S1  (round 4) BASELINE config 2: 1,000,000 reads x 150 bp from a 1 Mbp genome (seeds 1/2), k = 21, -c 10 -s 100
      (159 MB of text: 2 chunks).
  S3full (round 4) BASELINE config 5 whole: 50,000,000 reads (G = 50 Mbp, seeds 6/7), k = 63, -c 10 -s 100: all 78 Chunker
      chunks through the REFERENCE find_kmers(chunk, 63, 10) (~10 GB of dict per process: --procs 4, ~75 minutes).
Recorded per table: rows, sum, sha256 of the TSV text ("k-mer\t{base}_Count\n" + sorted rows), sha256 of the
concatenated keys and of the little-endian u64 counts; plus the chunk offsets.  Nothing of the reference's
source is stored.
"""
import argparse
import hashlib
import importlib.util
import json
import multiprocessing as mp
import os
import shutil
import sys
import time
from pathlib import Path

sys.dont_write_bytecode = True
REF = Path("/root/reference")
HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
COMP = str.maketrans("ACGT", "TGCA")


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, REF / rel)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def count_chunk(args):
    """One process = one countKmers task (bin/mercat2.py:112-114)."""
    path, k, c, want_canonical, check_direct = args
    kmers = load("ref_kmers", "lib/mercat2_kmers.py")
    t0 = time.time()
    if check_direct == "only":  # the reference's own filter, nothing else (the big samples: no second dict)
        fwd = kmers.find_kmers(Path(path), k, c)
        os.unlink(path)  # (the chunk file is not needed again: keep the disk use of an 8 GB sample bounded)
        return fwd, None, {"path": os.path.basename(path), "survivors": len(fwd), "seconds": round(time.time() - t0, 1)}
    raw = kmers.find_kmers(Path(path), k, 1)
    fwd = {key: n for key, n in raw.items() if n >= c}
    if check_direct:
        assert fwd == kmers.find_kmers(Path(path), k, c), "filter differs from find_kmers(min_count=c)"
    can = None
    if want_canonical:
        folded = {}
        get = folded.get
        for key, n in raw.items():
            rc = key.translate(COMP)[::-1]
            if rc < key:
                key = rc
            folded[key] = get(key, 0) + n
        can = {key: n for key, n in folded.items() if n >= c}
    info = {"path": os.path.basename(path), "distinct": len(raw), "windows": int(sum(raw.values())),
            "survivors": len(fwd), "seconds": round(time.time() - t0, 1)}
    return fwd, can, info


def digest(base, table):
    import numpy as np
    keys = sorted(table)
    h = hashlib.sha256()
    h.update(("k-mer\t%s_Count\n" % base).encode())
    step = 1 << 18
    for a in range(0, len(keys), step):
        h.update("".join("%s\t%d\n" % (x, table[x]) for x in keys[a:a + step]).encode())
    hk = hashlib.sha256()
    for a in range(0, len(keys), step):
        hk.update("".join(keys[a:a + step]).encode())
    counts = np.array([table[x] for x in keys], dtype="<u8")
    return {"rows": len(keys), "sum": int(counts.sum()), "sha256": h.hexdigest(), "keys_sha256": hk.hexdigest(),
            "counts_sha256": hashlib.sha256(counts.tobytes()).hexdigest(), "max_count": int(counts.max()) if len(keys) else 0}


def run_sample(tag, base, genome, gseed, reads, rseed, k, c, procs, want_canonical, first_chunks, tmp, direct_only=False):
    from mercat2_amd import native
    from mercat2_amd.chunker import chunk_offsets
    chunker = load("ref_chunker", "lib/mercat2_Chunker.py")
    work = Path(tmp) / tag
    shutil.rmtree(work, ignore_errors=True)
    (work / "chunks").mkdir(parents=True)
    data = native.synth_reads(genome, gseed, reads, 150, rseed, 0, 0)
    fna = work / (base + ".fna")
    with open(fna, "wb") as f:
        f.write(memoryview(data))
    ours = chunk_offsets(data, 100 << 20)
    del data
    ch = chunker.Chunker(str(fna), str(work / "chunks"), "100M", ">")
    files = sorted(ch.files)
    offs, pos = [], 0
    for p in files:
        offs.append(pos)
        pos += os.path.getsize(p)
    offs.append(pos)
    assert offs == list(ours), "the product's cut points differ from the reference Chunker's"
    if first_chunks:
        files = files[:first_chunks]
    print(tag, "chunks", len(files), "of", len(offs) - 1, flush=True)
    jobs = [(p, k, c, want_canonical, "only" if direct_only else (i == len(files) - 1 and not first_chunks)) for i, p in enumerate(files)]
    if direct_only:
        os.unlink(fna)
    fwd_total, can_total, infos = {}, {}, []
    with mp.get_context("fork").Pool(procs, maxtasksperchild=1) as pool:
        for fwd, can, info in pool.imap(count_chunk, jobs):
            print(tag, info, flush=True)
            infos.append(info)
            for key, n in fwd.items():  # the dict sum of bin/mercat2.py:123-127
                fwd_total[key] = fwd_total.get(key, 0) + n
            if can is not None:
                for key, n in can.items():
                    can_total[key] = can_total.get(key, 0) + n
    out = {"genome": genome, "genome_seed": gseed, "reads": reads, "read_seed": rseed, "read_len": 150, "k": k, "c": c,
           "chunk_mib": 100, "basename": base, "offsets": offs[:len(files) + 1], "chunks_used": len(files),
           "chunks_total": len(offs) - 1, "per_chunk": infos, "forward": digest(base, fwd_total)}
    if want_canonical:
        out["canonical"] = digest(base, can_total)
    shutil.rmtree(work, ignore_errors=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=5)
    ap.add_argument("--only", choices=("s2", "s3", "s1", "s3full"), default=None)
    ap.add_argument("--tmp", default="/tmp/mk_s2_golden")
    args = ap.parse_args()
    if not REF.is_dir():
        sys.exit("needs /root/reference")
    dst = HERE / "expected_s2.json"
    res = json.loads(dst.read_text()) if dst.exists() else {}
    if args.only in (None, "s2"):
        res["S2|k31|c10|s100"] = run_sample("s2", "S2", 10_000_000, 3, 10_000_000, 4, 31, 10, args.procs, True, 0, args.tmp)
        dst.write_text(json.dumps(res, indent=1, sort_keys=True))
    if args.only in (None, "s3"):
        # 1.4 M reads hold the first two 100 MiB chunks of S3 (a read is ~161 bytes of text); the third "chunk" here is
        # the cut-off remainder and is not used
        res["S3head|k63|c2|s100|chunks2"] = run_sample("s3", "S3", 50_000_000, 6, 1_400_000, 7, 63, 2, min(args.procs, 2), False, 2, args.tmp)
        dst.write_text(json.dumps(res, indent=1, sort_keys=True))
    if args.only == "s1":
        res["S1|k21|c10|s100"] = run_sample("s1", "S1", 1_000_000, 1, 1_000_000, 2, 21, 10, min(args.procs, 2), False, 0, args.tmp)
        dst.write_text(json.dumps(res, indent=1, sort_keys=True))
    if args.only == "s3full":
        res["S3|k63|c10|s100"] = run_sample("s3full", "S3", 50_000_000, 6, 50_000_000, 7, 63, 10, min(args.procs, 4), False, 0, args.tmp, direct_only=True)
        dst.write_text(json.dumps(res, indent=1, sort_keys=True))
    print(json.dumps({k: {m: v[m]["rows"] for m in ("forward", "canonical") if m in v} for k, v in res.items()}))


if __name__ == "__main__":
    main()
