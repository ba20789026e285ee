#!/usr/bin/env python3
"""bench.py -- headline benchmark: bases/s of the k-mer counting hot path on MI355X.

Workload (BASELINE.json metric: "bases/sec at k=31, 10M x 150bp"; SURVEY.md 8d "S2"):
  10,000,000 reads x 150 bp sampled (both strands) from a 10 Mbp iid genome (seeds 3/4),
  FASTA text ~1.6 GB, k=31, forward-strand keys (reference behaviour), -c 10, -s 100:
  the reference's Chunker cut points (>= 100 MiB chunks, 16 of them) are the filter units.
One "step" = the whole sample once: for every chunk  raw FASTA bytes (already resident in HBM)
  -> GPU parse -> 2-bit pack -> count -> keep count >= 10 -> add into the running table,
  then the sorted (key,count) export of the merged table on the device (+ for N > 1 the
  key-range all-to-all merge across ranks over RCCL).  value = bases of all ranks / time.

N > 1, one process per GPU (the reference's analogue is the Ray fan-out of chunks,
bin/mercat2.py:119-127,336-339):
  --scaling weak   (default) every rank counts its own 10M reads (same genome, disjoint read
                   indices), then the ranks merge their tables;
  --scaling strong the 16 chunks of the ONE S2 sample are dealt chunk i -> rank i mod N,
                   every rank filters its own chunks, then the ranks merge.
  The other mode is timed after the main region and reported under "also".
  Launched by torch.distributed.run (RANK/WORLD_SIZE in the environment) or by itself: with
  --gpus N > 1 and no RANK set, this process starts N children (before anything here touches the
  GPU) and prints rank 0's line.  "ranks_seen" is an all_reduce of ones, "rccl" says the backend.

Extra objects on the JSON line:
  roofline     dominant kernel = the LDS count kernel; achieved = algorithmic bytes per launch
               (windows * 16 B, k <= 32: one 8-byte key compare + one 8-byte count read-modify-write
               per window; 24 B for two-word keys; SURVEY.md 8d / DESIGN.md) / mean launch time
               measured with HIP events on the engine's stream; peak = 8 TB/s HBM3E.  `stage_*`
               repeat the calculation for the whole counting stage (partition + count kernels, + the
               0.25 B per symbol packed read).
  cpu_baseline the CPU oracle (Python restatement of the reference, oracle/cpu_ref.py) timed on
               this box's host cores on a bounded sample of the same reads (one real S2 chunk
               split over the processes).
  file_to_tsv  the reference's `Time to count` window (bin/mercat2.py:335-346): the S2 sample as a
               file in the page cache -> TSV closed, through harness.run_sample; plain and .gz.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

GENOME, GENOME_SEED, READS, READ_LEN, READ_SEED = 10_000_000, 3, 10_000_000, 150, 4
K, MIN_COUNT, CHUNK_MIB = 31, 10, 100
HBM_PEAK_GBS = 8000.0


# ------------------------------------------------------------------------------ CPU baseline
def cpu_sample_worker(args):
    """One process: its slice of a real S2 chunk (reads of the 10 Mbp genome), counted by the oracle."""
    from oracle import cpu_ref
    from mercat2_amd import native
    genome, gseed, rseed, first, reads, k, c = args
    data = native.synth_reads(genome, gseed, reads, READ_LEN, rseed, 0, first).tobytes()
    t0 = time.perf_counter()
    table = cpu_ref.count_text(data, k, c)
    return reads * READ_LEN, time.perf_counter() - t0, len(table)


def cpu_baseline(k, c, genome, gseed, rseed, chunk_reads):
    """Bounded CPU run of the oracle on the benchmark's own reads: the first chunk of the sample
    (chunk_reads reads) is split into P equal slices, one per process; every process counts its slice
    as find_kmers would (dict of strings, per-file filter).  The slices' dicts (~4.5 M keys each at
    S2) are far out of cache, as the real chunk's 20 M keys are; smaller dicts are, if anything, kind
    to the CPU."""
    import multiprocessing as mp
    cores = max(1, min(os.cpu_count() or 1, 16))
    per = max(1000, chunk_reads // 16)  # the same slice size whatever the core count (~0.5 GB of dict per process at S2)
    rounds = 3                          # slices per process, one after the other: ~10-30 s of CPU work per core
    jobs = [(genome, gseed, rseed, i * per, per, k, c) for i in range(cores * rounds)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(cpu_sample_worker, jobs)
    wall = time.perf_counter() - t0
    bases = sum(r[0] for r in res)
    busy = sum(r[1] for r in res)
    out = {"value": bases / wall, "unit": "bases/s", "cores": cores, "kind": "port",
           "sample": "reads 0..%d of the benchmark sample (genome %d bp, seeds %d/%d: the head of its first %d-read chunk), "
                     "%d slices of %d reads x %d bp, %d processes, k=%d, c=%d, pure-Python oracle (oracle/cpu_ref.py); "
                     "per-core rate %.3g bases/s"
                     % (per * len(jobs), genome, gseed, rseed, chunk_reads, len(jobs), per, READ_LEN, cores, k, c, bases / busy)}
    cal = ROOT / "profiles" / "round2_calibration.json"
    if cal.exists():
        j = json.loads(cal.read_text())
        out["calibration_ratio"] = j.get("ratio_oracle_over_reference")
        out["calibration_note"] = "oracle rate / reference find_kmers rate on the same sample, measured in the build " \
                                  "container by tools/calibrate_cpu_ref.py (profiles/round2_calibration.json)"
    return out


# ------------------------------------------------------------------------------ self-launch
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """Start n copies of this script, one per GPU, and relay rank 0's line.  This (parent) process
    never touches the GPU and execs nothing: the children are fresh processes."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MK_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    # exactly one line on stdout: rank 0's JSON (whatever else a library printed there goes to stderr)
    lines = out.decode().splitlines()
    for ln in lines:
        (sys.stdout if ln.startswith("{") else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc]
    if bad:
        raise SystemExit("bench.py: rank(s) failed: %s" % bad)


# ----------------------------------------------------------------------------------- one rank
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak = every rank its own sample of --reads reads; strong = the chunks of ONE sample dealt i mod N")
    ap.add_argument("--reads", type=int, default=READS, help="reads per sample (default: the BASELINE workload)")
    ap.add_argument("--k", type=int, default=K)
    ap.add_argument("--genome", type=int, default=GENOME, help="genome length of the synthetic sample")
    ap.add_argument("--genome-seed", type=int, default=GENOME_SEED)
    ap.add_argument("--read-seed", type=int, default=READ_SEED)
    ap.add_argument("--sub-ppm", type=int, default=0, help="per-base substitution rate, parts per million (S2e: 10000)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-file-leg", action="store_true", help="skip the file-to-TSV leg")
    ap.add_argument("--no-also", action="store_true", help="N > 1: do not time the other scaling mode after the main region")
    ap.add_argument("--canonical", action="store_true",
                    help="count min(kmer, reverse complement) (BASELINE config 3 names it; an opt-in extension, not the "
                         "reference's forward-strand behaviour -- the default run keeps that)")
    ap.add_argument("--contexts", type=int, default=int(os.environ.get("MK_BENCH_CONTEXTS", "0")),
                    help="engine contexts (HIP streams) per GPU; chunks are dealt round-robin and counted "
                         "concurrently, the tables are merged on the device at the end of the step")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "RANK" not in os.environ and args.gpus > 1:
        # no launcher: be one (before torch.cuda / HIP is touched in this process)
        return launch_ranks(args.gpus, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    backend = os.environ.get("MK_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of the N>1 path on a 1-GPU box
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit("%d ranks but %d GPU(s): RCCL needs one device per rank (MK_BENCH_BACKEND=gloo rehearses on fewer)" % (world, ndev))
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    from mercat2_amd import native
    from mercat2_amd.chunker import chunk_offsets
    from mercat2_amd.dist import merge_ranks
    from concurrent.futures import ThreadPoolExecutor

    k = args.k
    canonical = bool(args.canonical)
    nctx = args.contexts if args.contexts > 0 else native.default_streams(k, native.ALPHABET_NT2)
    ctxs = [native.Counter(k, native.ALPHABET_NT2, device=local, canonical=canonical) for _ in range(nctx)]
    ctx = ctxs[0]
    pool = ThreadPoolExecutor(nctx) if nctx > 1 else None
    key_bits = 2 * k
    out_cap = (2 * args.genome + 1024) * (1 if args.sub_ppm == 0 else 12)  # distinct forward-strand k-mers of both strands, upper bound
    words = ctx.words_per_key()
    out_keys = torch.empty(out_cap * words, dtype=torch.int64, device=dev)
    out_cnts = torch.empty(out_cap, dtype=torch.int64, device=dev)

    # ---- synthetic input: generate on the host, cut like the reference Chunker, move to HBM
    def load_sample(mode):
        """(device chunk tensors of this rank, chunks of the sample, bases this step counts over all ranks, gen seconds)."""
        t0 = time.perf_counter()
        first = rank * args.reads if mode == "weak" else 0
        host = native.synth_reads(args.genome, args.genome_seed, args.reads, READ_LEN, args.read_seed, args.sub_ppm, first)
        offs = chunk_offsets(host, CHUNK_MIB * 1024 * 1024) if host.nbytes >= CHUNK_MIB * 1024 * 1024 else [0, host.nbytes]
        spans = list(zip(offs[:-1], offs[1:]))
        mine = spans if mode == "weak" else spans[rank::world]
        if mode == "weak":
            whole = torch.from_numpy(host).to(dev)
            parts = [whole[a:b] for a, b in mine]
        else:
            parts = [torch.from_numpy(host[a:b]).to(dev) for a, b in mine]
        torch.cuda.synchronize()
        del host
        total = args.reads * READ_LEN * (world if mode == "weak" else 1)
        return parts, len(spans), total, time.perf_counter() - t0

    def make_step(parts):
        ptrs = [(p.data_ptr(), p.numel()) for p in parts]

        def count_share(i):
            c = ctxs[i]
            c.reset()
            for ptr, n in ptrs[i::nctx]:  # every chunk is filtered on its own (the per-chunk -c rule)
                c.count_device(ptr, n, MIN_COUNT)

        def step():
            if pool is None:
                count_share(0)
            else:
                list(pool.map(count_share, range(nctx)))
                for c in ctxs[1:]:  # sum the other contexts' survivors into context 0, on the device
                    ctx.merge_from(c)
            if world > 1:
                merge_ranks(ctx, key_bits, device=dev)
            return ctx.export_pairs_device(out_keys.data_ptr(), out_cnts.data_ptr(), out_cap)
        return step

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, steps, warmup, profile):
        for _ in range(warmup):
            step()
        if profile:
            for c in ctxs:
                c.reset_stats()
                c.set_profiling(True)
        fence()
        t0 = time.perf_counter()
        rows = 0
        for _ in range(steps):
            rows = step()
        fence()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        r = torch.tensor([rows, 1], dtype=torch.int64, device=red_dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(r, op=dist.ReduceOp.SUM)
        return float(t.item()), int(r[0].item()), int(r[1].item())

    mode = args.scaling if world > 1 else "weak"
    parts, nchunks, total_bases, gen_s = load_sample(mode)
    step = make_step(parts)
    dt, total_rows, ranks_seen = timed(step, args.steps, args.warmup, True)
    stats = [c.stats() for c in ctxs]
    for c in ctxs:
        c.set_profiling(False)
    st = dict(stats[0])
    for other in stats[1:]:  # totals over the contexts of this GPU
        for key, val in other.items():
            if key.startswith(("ms_", "n_")) or key in ("windows", "exotic_windows", "symbols", "raw_bytes", "chunks", "records", "distinct"):
                st[key] += val

    # After the timed region: the same kernels alone on the GPU (one context, one pass), so that the
    # dominant kernel's duration can also be read without the other stream's kernels on its CUs.
    solo = None
    if nctx > 1 and rank == 0:
        c = ctxs[0]
        c.reset()
        c.reset_stats()
        c.set_profiling(True)
        for p in parts:
            c.count_device(p.data_ptr(), p.numel(), MIN_COUNT)
        torch.cuda.synchronize()
        solo = c.stats()
        c.set_profiling(False)

    # the other scaling mode, timed after the main region (N > 1 only)
    also = None
    if world > 1 and not args.no_also:
        other_mode = "strong" if mode == "weak" else "weak"
        parts2, nchunks2, total2, _ = load_sample(other_mode)
        dt2, rows2, _ = timed(make_step(parts2), args.steps, 1, False)
        also = {"scaling": other_mode, "value": total2 * args.steps / dt2, "unit": "bases/s", "ms_per_step": dt2 / args.steps * 1e3,
                "rows": rows2, "chunks": nchunks2, "note": "timed after the main region, same steps, 1 warm-up"}
        del parts2

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = total_bases * args.steps / dt
        two_word = st["mode_name"] == "hash128"
        launches = max(1, st["n_count"])
        ms_launch = st["ms_count"] / launches
        packed_windows = st["windows"] - st["exotic_windows"]
        bytes_per_window = 24 if two_word else 16
        alg_bytes = packed_windows * bytes_per_window  # over all timed launches of the count kernel
        achieved = alg_bytes / launches / (ms_launch * 1e-3) / 1e9 if ms_launch > 0 else 0.0
        stage_ms = (st["ms_count"] + st["ms_part"]) / launches
        stage_bytes = alg_bytes + st["symbols"] * 0.25
        stage_achieved = stage_bytes / launches / (stage_ms * 1e-3) / 1e9 if stage_ms > 0 else 0.0
        kernel_name = {"hash64": "mk_sk_count_k" if 18 <= k <= 32 else "mk_part_count_k", "dense": "mk_count_dense_k",
                       "byref": "mk_count_byref_k", "hash128": "mk_sk2_count_k"}.get(st["mode_name"], "?")
        if os.environ.get("MK_NO_PARTITION"):
            kernel_name = "mk_count_hash64_k"
        # HBM bytes per launch of that kernel from the committed PMC passes (FETCH_SIZE / WRITE_SIZE in
        # separate rocprofv3 runs of this command, gfx950 correction applied: tools/trim_profiles.py)
        traffic, traffic_src = None, None
        for name in ("round2_pmc_traffic.json", "round1_pmc_traffic.json"):
            pmc_file = ROOT / "profiles" / name
            if pmc_file.exists() and args.reads == READS and k == K and not canonical:
                pmc = json.loads(pmc_file.read_text())
                if kernel_name in pmc:
                    traffic = pmc[kernel_name]["hbm_bytes_per_launch"]
                    traffic_src = "profiles/" + name.replace(".json", ".csv")
                    break
        is_s2 = (args.reads, args.genome, k, args.sub_ppm) == (READS, GENOME, K, 0)
        line = {
            "metric": "bases/sec at k=31, 10Mx150bp", "value": value, "unit": "bases/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": mode, "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "ranks_seen": ranks_seen, "rccl": bool(world > 1 and backend == "nccl"), "backend": backend if world > 1 else None,
            "config": {"workload": "%s: %d reads x %d bp %s from a %d bp genome, k=%d, -c %d, -s %d (%d chunks per sample), %s keys"
                                   % ("S2" if is_s2 else "custom", args.reads, READ_LEN,
                                      "per GPU" if mode == "weak" else "in all (one sample, chunks dealt i mod N)", args.genome, k,
                                      MIN_COUNT, CHUNK_MIB, nchunks, "canonical" if canonical else "forward-strand"),
                       "reads_per_gpu": args.reads if mode == "weak" else args.reads / world, "read_len": READ_LEN, "k": k,
                       "min_count": MIN_COUNT, "chunk_mib": CHUNK_MIB, "chunks": nchunks, "mode": st["mode_name"],
                       "contexts_per_gpu": nctx,
                       "parallelism": "chunks->ranks, key-range all-to-all merge" if world > 1 else "1 GPU"},
            "distinct_kmers_per_s": total_rows * args.steps / dt,
            "distinct_prefilter_per_s": st["distinct"] * world / dt,  # distinct keys per chunk, before the -c filter
            "rows": total_rows,
            "kernel_ms_per_step": {n: st["ms_" + n] / args.steps for n in ("parse", "pack", "part", "count", "exotic", "filter", "export")},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "launches": launches, "ms_per_launch": ms_launch,
                         "algorithmic_bytes_per_launch": alg_bytes / launches, "bytes_per_window": bytes_per_window,
                         "stage_ms_per_launch": stage_ms, "stage_achieved": stage_achieved,
                         "stage_frac": stage_achieved / HBM_PEAK_GBS},
            "input_gen_s": gen_s,
        }
        if also:
            line["also"] = also
        if solo and solo["n_count"] and solo["ms_count"] > 0:
            solo_ms = solo["ms_count"] / solo["n_count"]
            solo_ach = (solo["windows"] - solo["exotic_windows"]) * bytes_per_window / solo["n_count"] / (solo_ms * 1e-3) / 1e9
            # (not `achieved`: that one is measured inside the timed region, where two contexts share the GPU)
            line["roofline"]["one_context"] = {"ms_per_launch": solo_ms, "achieved": solo_ach, "frac": solo_ach / HBM_PEAK_GBS,
                                               "note": "same kernel, untimed extra pass with one context"}
        if rank == 0:
            # the box's own stream-copy rate beside the nominal peak (1 GiB device-to-device copy, read + write
            # bytes / time, best of 5, after the timed region)
            try:
                src = torch.empty(1 << 28, dtype=torch.int32, device=dev)
                dst = torch.empty_like(src)
                best = None
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    dst.copy_(src)
                    e1.record()
                    e1.synchronize()
                    ms = e0.elapsed_time(e1)
                    best = ms if best is None or ms < best else best
                line["roofline"]["measured_stream_copy"] = 2 * src.numel() * 4 / (best * 1e-3) / 1e9
                line["roofline"]["frac_of_measured_copy"] = achieved / line["roofline"]["measured_stream_copy"]
                del src, dst
            except Exception as e:  # (a probe, not the product: never fail the bench line over it)
                line["roofline"]["measured_stream_copy"] = None
        if world == 1 and not args.no_file_leg:
            line["file_to_tsv"] = file_to_tsv_leg(args, k, canonical)
        if not args.no_cpu and world == 1:
            chunk_reads = max(1, args.reads // max(1, nchunks))
            line["cpu_baseline"] = cpu_baseline(k, MIN_COUNT, args.genome, args.genome_seed, args.read_seed, chunk_reads)
            line["speedup_vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
            if "file_to_tsv" in line:  # the target is stated on the file-to-TSV window
                cpu = line["cpu_baseline"]["value"]
                line["file_to_tsv"]["plain_vs_cpu_baseline"] = line["file_to_tsv"]["plain_bases_per_s"] / cpu
                line["file_to_tsv"]["gz_vs_cpu_baseline"] = line["file_to_tsv"]["gz_bases_per_s"] / cpu
                line["file_to_tsv"]["vs_cpu_note"] = ("the CPU figure is counting only (text already in memory): against a "
                                                      "CPU run that also had to inflate the .gz the ratio would be larger")
        print(json.dumps(line))
        sys.stdout.flush()
    for c in ctxs:
        c.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def file_to_tsv_leg(args, k, canonical):
    """The reference's `Time to count` window on the same sample: file (page cache) -> TSV closed,
    through the product host layer (harness.run_sample = mk_count_file + mk_write_tsv).  Best of two
    runs each; the .gz is one ordinary DEFLATE stream (zlib level 1)."""
    import shutil
    import tempfile
    import zlib
    from mercat2_amd import harness, native
    d = tempfile.mkdtemp(prefix="mk_bench_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        data = native.synth_reads(args.genome, args.genome_seed, args.reads, READ_LEN, args.read_seed, args.sub_ppm, 0)
        plain = os.path.join(d, "S2.fna")
        with open(plain, "wb") as f:
            f.write(memoryview(data))
        gz = os.path.join(d, "S2.fna.gz")
        t0 = time.perf_counter()
        co = zlib.compressobj(1, zlib.DEFLATED, 31)
        with open(gz, "wb") as f:
            mv = memoryview(data)
            for a in range(0, len(mv), 64 << 20):
                f.write(co.compress(mv[a:a + (64 << 20)]))
            f.write(co.flush())
        gzip_s = time.perf_counter() - t0
        del data
        res = {}
        for name, path in (("plain", plain), ("gz", gz)):
            best, st_best, rows = None, None, 0
            for _ in range(2):
                st = {}
                out = os.path.join(d, "S2_counts.tsv")
                lines = []
                t0 = time.perf_counter()
                harness.run_sample("S2", path, out, k, MIN_COUNT, CHUNK_MIB, canonical=canonical, stats=st, report=lines.append)
                dt = time.perf_counter() - t0
                if best is None or dt < best:
                    best, st_best = dt, st
                rows = int(lines[0].split(":")[1]) if lines and ":" in lines[0] else 0
            res[name + "_s"] = best
            res[name + "_bases_per_s"] = args.reads * READ_LEN / best
            res[name + "_threads"] = st_best.get("threads")
            res[name + "_file_bytes"] = os.path.getsize(path)
            res["rows"] = rows
        res["bases_per_s"] = res["plain_bases_per_s"]
        res["threads"] = res["plain_threads"]
        res["note"] = "file in the page cache -> TSV closed (read/inflate + chunk + count + sort + write); best of 2; " \
                      "gz = one DEFLATE stream, zlib level 1 (made in %.1f s)" % gzip_s
        return res
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
