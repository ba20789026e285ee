"""Command line of the counting path, flag-compatible with bin/mercat2.py (lines 41-50, 68-79,
207-215, 253-283, 312-346, 411-448 of the reference): -i/-f/-k/-n/-c/-s/-o/-replace.

The count phase and what stands directly in front of and behind it: inputs are
FASTA (nucleotide or protein; plain or .gz); nucleotide inputs go through removeN first unless
-skipclean is given (bin/mercat2.py:239-244), exactly as in the reference; the combined tables are
written after counting.  FASTQ conversion and QC, ORF calling (-prod / -fgs), reports, PCA and plots
belong to the reference's other layers: their flags are accepted where they change nothing here
(-lowmem, -pca, -debug, -category_file) and refused with a clear message where the run would need
that layer's output (-prod, -fgs, FASTQ input).
"""
from __future__ import annotations

import argparse
import os
import shutil
import sys
import timeit
from pathlib import Path

from . import __version__
from .fasta import removeN_background, removeN_text
from .kmers import read_fasta_bytes
from .harness import run_raw_clean, run_sample, run_text
from .report import merge_counters, merge_counters_T

FILE_EXT_FASTQ = [".fq", ".fastq", ".fq.gz", ".fastq.gz"]

FILE_EXT_NUCLEOTIDE = [".fasta", ".fa", ".fna", ".ffn", ".fasta.gz", ".fa.gz", ".fna.gz", ".ffn.gz"]
FILE_EXT_PROTEIN = [".faa", ".faa.gz"]


def parseargs(argv=None):
    p = argparse.ArgumentParser(description="MerCat2 k-mer counting on MI355X (count phase only)")
    p.add_argument("-i", required=False, default=list(), help="path to input file", nargs="+")
    p.add_argument("-f", type=str, required=False, help="path to folder containing input files")
    p.add_argument("-k", type=int, required=True, help="kmer length")
    p.add_argument("-n", type=int, default=os.cpu_count() or 1,
                   help="no of cores [auto detect]: samples read (inflated) and counted concurrently, at most 8")
    p.add_argument("-c", type=int, default=10, help="minimum kmer count [10]")
    p.add_argument("-s", type=int, default=100, required=False, help="Split into x MB files. [100]")
    p.add_argument("-o", type=str, default="mercat_results", required=False, help="Output folder")
    p.add_argument("-replace", action="store_true", help="Replace existing output directory [False]")
    p.add_argument("-skipclean", action="store_true", help="skip trimming of the sequences [False]")
    p.add_argument("-toupper", action="store_true", help="convert all input sequences to uppercase [False]")
    # flags of the reference's other layers (bin/mercat2.py:45-58)
    p.add_argument("-prod", action="store_true", help="(MerCat2: ORF calling with prodigal) not part of this engine")
    p.add_argument("-fgs", action="store_true", help="(MerCat2: ORF calling with FragGeneScanRs) not part of this engine")
    p.add_argument("-lowmem", action="store_true", help="(MerCat2: incremental PCA) accepted, no effect: no PCA here")
    p.add_argument("-pca", action="store_true", help="(MerCat2: PCA plots) accepted, no effect")
    p.add_argument("-debug", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("-category_file", type=str, required=False, help=argparse.SUPPRESS)
    p.add_argument("-gpus", type=int, default=None,
                   help="number of GPUs to use, devices 0..N-1 [all visible]: a sample that is chunked (-s) has its chunks spread "
                        "over them (chunk i on GPU i mod N, tables summed by peer copies); small samples go one per GPU")
    p.add_argument("-gpu", type=int, default=None, help="use exactly this one HIP device (overrides -gpus)")
    p.add_argument("-streams", type=int, default=None,
                   help="engine contexts counting chunks concurrently [2 for one-word keys, else 1]")
    p.add_argument("-union", action="store_true",
                   help="combined_<type>.tsv as the true union of the samples' tables (one row per k-mer in any sample). Default: "
                        "the rows MerCat2's merge_tsv writes, whose streaming loop leaves out or misplaces k-mers that not "
                        "all samples share (combined_<type>_T.tsv is always the union)")
    p.add_argument("-canonical", action="store_true",
                   help="EXTENSION (not MerCat2 behaviour): count min(kmer, reverse complement) for nucleotide input")
    p.add_argument("--version", "-v", action="version", version=f"mercat2_amd {__version__}")
    args = p.parse_args(argv)
    if not args.i and not args.f:
        p.error("Please provide either an input file (-i) or an input folder (-f)")
    for filename in args.i:
        if not os.path.isfile(filename):
            p.error(f"file '{filename}' is not valid.\n")
    if args.f and not os.path.isdir(args.f):
        p.error(f"folder {args.f} is not valid.\n")
    if args.prod or args.fgs:
        p.error("-prod / -fgs call ORFs with prodigal / FragGeneScanRs before counting amino-acid k-mers; that layer is "
                "not part of this engine: run the ORF caller and pass its .faa output with -i / -f")
    return args, p


def classify(path: Path):
    """(type, basename) by the reference's extension tables (bin/mercat2.py:26-28, 264-283)."""
    suffixes = path.suffixes
    ext = ""
    for i in reversed(range(len(suffixes))):
        cand = "".join(suffixes[i:])
        if cand in FILE_EXT_NUCLEOTIDE + FILE_EXT_PROTEIN:
            ext = cand
    if not ext:
        for i in reversed(range(len(suffixes))):
            if "".join(suffixes[i:]) in FILE_EXT_FASTQ:
                raise SystemExit(f"'{path.name}': FASTQ input needs MerCat2's fastq_to_fasta layer (fastp / fastqc), which is "
                                 "not part of this engine; convert it to FASTA first")
        return None, None
    base = path.name[: -len(ext)]
    return ("protein" if ext in FILE_EXT_PROTEIN else "nucleotide"), base


def main(argv=None) -> int:
    args, parser = parseargs(argv)
    out = Path(args.o)
    if out.exists():
        if args.replace:
            shutil.rmtree(out)
        else:
            parser.error(f"Output folder exists, please specify another folder or use the flag '-replace' to override the files. '{out}'")
    out.mkdir(0o777, True, True)
    from . import native
    visible = native.device_count()
    if visible < 1:
        raise SystemExit("mercat2_amd: no HIP device is visible (this engine has no CPU fallback)")
    if args.gpu is not None:
        if not 0 <= args.gpu < visible:
            parser.error(f"-gpu {args.gpu}: {visible} device(s) visible")
        devices = [args.gpu]
    else:
        want = visible if args.gpus is None else args.gpus
        if not 1 <= want <= visible:
            parser.error(f"-gpus {want}: {visible} device(s) visible")
        devices = list(range(want))
    print(f"\nStarting mercat2_amd v{__version__} with k-mer {args.k} on GPU{'s' if len(devices) > 1 else ''} "
          f"{','.join(str(d) for d in devices)}\n")
    files = [Path(f) for f in args.i]
    if args.f:
        folder = Path(os.path.abspath(os.path.expanduser(args.f)))
        files += [folder / name for name in sorted(os.listdir(folder)) if (folder / name).is_file()]
    samples = {"nucleotide": {}, "protein": {}}
    for f in files:
        kind, base = classify(f.expanduser().absolute())
        if kind:
            samples[kind][base] = f

    from concurrent.futures import ThreadPoolExecutor
    # ---- "Loading files" (bin/mercat2.py:229-298): nucleotide FASTA goes through removeN unless -skipclean.  The text
    # rewrite is native and fast; the level-9 gzip of <base>_clean.fna.gz is not (~1.5 MB/s), so the files are written
    # by background threads while the counting below already runs on the cleaned text in memory.
    print("Loading files")
    load_start = timeit.default_timer()
    clean = not args.skipclean
    cleaned = {}  # base -> [clean file, raw bytes (until counted), future of (.gz size, stats), holder of the cleaned text, timings]
    gz_writers = ThreadPoolExecutor(max(1, min(int(args.n), 16)))
    if clean and samples["nucleotide"]:
        import threading
        budget = [8 << 30]  # bytes of input held in memory between loading and counting; samples beyond it are read when counted
        budget_lock = threading.Lock()

        def load(item):
            base, f = item
            t = {}
            with budget_lock:
                room = budget[0] > 0
                budget[0] -= os.stat(f).st_size * (4 if str(f).endswith(".gz") else 1)
            if not room:
                return base, None
            t0 = timeit.default_timer()
            raw = read_fasta_bytes(f)
            t["read_s"] = timeit.default_timer() - t0
            path, fut, holder = removeN_background(f, raw, out / "clean", args.toupper, gz_writers, timings=t,
                                                   limit=args.s * 1024 * 1024)
            return base, [path, raw, fut, holder, t]
        with ThreadPoolExecutor(max(1, min(int(args.n), 8, len(samples["nucleotide"])))) as pool:
            for base, job in pool.map(load, samples["nucleotide"].items()):
                cleaned[base] = job
    print(f"Time to load {len(samples['nucleotide']) + len(samples['protein'])} files: {round(timeit.default_timer() - load_start, 2)} seconds")

    for kind in ("nucleotide", "protein"):
        if not samples[kind]:
            continue
        print("Processing Nucleotides" if kind == "nucleotide" else "Processing protein")
        tsv_dir = out / f"tsv_{kind}"
        tsv_dir.mkdir(parents=True, exist_ok=True)
        start = timeit.default_timer()
        # Samples are independent (bin/mercat2.py:336-339).  A '.gz' sample is bound by its one inflating
        # thread, so up to -n samples (at most 8) are in flight at once, each with its own contexts; the
        # lines the reference prints per sample are kept and shown in sample order.
        tables = {}  # sample -> its table, kept on the GPU for the combined table

        workers = max(1, min(int(args.n), max(8, 2 * len(devices)), len(samples[kind])))
        threads = max(2, 16 // workers)  # reader/decoder threads per sample: about 16 in all

        def one(numbered):
            idx, (base, f) = numbered
            lines = []
            # Several GPUs: a sample that is chunked spreads its chunks over all of them (run_sample /
            # run_text deal chunk i to GPU i mod N); a sample that is one chunk stays on ONE GPU, the samples taking
            # the GPUs in turn (no exchange at all) -- the reference's one Ray task per sample (bin/mercat2.py:336-339)
            home = devices[idx % len(devices)]
            t = {}
            t0 = timeit.default_timer()
            tsv = tsv_dir / f"{base}_counts.tsv"
            if kind == "nucleotide" and clean:
                limit = args.s * 1024 * 1024
                if cleaned[base] is None:  # (not loaded up front: the rewrite, its file, then the count, one after the other)
                    clean_file, _gc, text = removeN_text(f, out / "clean", args.toupper, timings=t)
                    chunked = args.s > 0 and os.stat(clean_file).st_size >= limit
                    run_text(base, text, tsv, args.k, args.c, args.s, chunked, device=home, devices=devices if chunked else None,
                             streams=args.streams, canonical=args.canonical, report=lines.append, keep=tables, timings=t)
                    del text
                    t_load = {}
                else:
                    clean_file, raw, fut, holder, t_load = cleaned[base]
                    # the table straight from the raw text, counted as removeN leaves it (N runs cut records: the GPU finds
                    # them in the parser's pass), while the rewrite and the level-9 gzip of the clean file run in the background
                    done = run_raw_clean(base, raw, tsv, args.k, args.c, args.toupper, limit, device=home, canonical=args.canonical,
                                         report=lines.append, keep=tables, timings=t)
                    cleaned[base][1] = raw = None
                    if done is not None:
                        holder["drop"]()
                    else:
                        # the sample may be chunked (the reference cuts the CLEANED file, and its size on disk decides,
                        # bin/mercat2.py:101, 243), or holds text whose rewrite the GPU does not reproduce: count the text
                        # the host rewrite produced (native, memory speed) -- without waiting for the level-9 gzip of the
                        # clean file: a DEFLATE stream only grows, so "chunked" is known the moment its bytes pass -s MiB.
                        holder["ready"].wait()
                        if "text" not in holder:
                            fut.result()  # (the rewrite failed: its exception surfaces here, as it would in MerCat2)
                        text = holder.pop("text")
                        decision = holder["decision"]
                        kw = dict(device=home, streams=args.streams, canonical=args.canonical, timings=t)
                        # the .gz cannot be larger than the text plus the stored-block overhead zlib falls back to
                        certain_whole = args.s <= 0 or len(text) + len(text) // 1000 + 4096 < limit
                        chunked = False if certain_whole else decision.wait(0)
                        if chunked is not None:
                            run_text(base, text, tsv, args.k, args.c, args.s, chunked, devices=devices if chunked else None,
                                     report=lines.append, keep=tables, **kw)
                        else:
                            # not known yet: count BOTH tables now (milliseconds each), publish the one the size selects
                            t_wait = timeit.default_timer()
                            cand = {}
                            for name, ch in (("whole", False), ("chunked", True)):
                                lc, kc = [], {}
                                run_text(base, text, str(tsv) + "." + name, args.k, args.c, args.s, ch, devices=devices if ch else None,
                                         report=lc.append, keep=kc, **kw)
                                cand[ch] = (str(tsv) + "." + name, lc, kc)
                            chunked = decision.wait()
                            if chunked is None:
                                fut.result()  # (the gzip writer failed)
                            t["decide_s"] = timeit.default_timer() - t_wait
                            path_, lc, kc = cand[bool(chunked)]
                            if os.path.exists(path_):
                                os.replace(path_, tsv)
                            lines.extend(lc)
                            tables.update(kc)
                            other_path, _, other_keep = cand[not chunked]
                            if os.path.exists(other_path):
                                os.unlink(other_path)
                            for ctx_ in other_keep.values():
                                ctx_.close()
                        del text
                t.update(t_load)
            else:
                run_sample(base, f, tsv, args.k, args.c, args.s, device=home,
                           devices=[home] + [d for d in devices if d != home], streams=args.streams, canonical=args.canonical,
                           report=lines.append, keep=tables, threads=threads if workers > 1 else 0, timings=t)
            if args.debug:
                t["total_s"] = timeit.default_timer() - t0
                lines.append(f"[debug] {base}: " + " ".join(f"{k_}={v:.3f}" if isinstance(v, float) else f"{k_}={v}" for k_, v in sorted(t.items())))
            return lines
        if workers == 1:
            results = map(one, enumerate(samples[kind].items()))
        else:
            pool = ThreadPoolExecutor(workers)
            results = pool.map(one, enumerate(samples[kind].items()))
        for lines in results:
            for line in lines:
                print(line)
        print(f"Time to count {args.k}-mers: {round(timeit.default_timer() - start, 2)} seconds")
        # combined_<type>.tsv: what createFigures writes first (bin/mercat2.py:146-150, merge_tsv), here
        # straight from the tables; samples without significant k-mers are left out, as there
        try:
            if tables:
                stem = "combined_Nucleotide" if kind == "nucleotide" else "combined_protein"
                rows = merge_counters(tables, out / (stem + ".tsv"), as_reference=not args.union)
                union_rows = merge_counters_T(tables, out / (stem + "_T.tsv"))  # bin/mercat2.py:154-157, read by beta diversity
                if rows != union_rows:
                    print(f"Note: {stem}.tsv has {rows} rows, the samples hold {union_rows} different k-mers: MerCat2's merge_tsv "
                          f"leaves out or misplaces k-mers that not all samples share; {stem}_T.tsv is the full table, and "
                          f"-union writes {stem}.tsv that way too")
        finally:
            for t in tables.values():
                t.close()
    # the clean files must be complete before the run ends
    wait_start = timeit.default_timer()
    for base, job in cleaned.items():
        if job is not None:
            job[2].result()  # (an error of the rewrite -- e.g. a record to be split without a name: IndexError, as in MerCat2 -- surfaces here)
    gz_writers.shutdown()
    if cleaned and args.debug:
        print(f"[debug] waited {round(timeit.default_timer() - wait_start, 2)} s more for the clean/*.fna.gz writers (gzip level 9)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
