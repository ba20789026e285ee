#!/usr/bin/env python3
"""bench.py -- headline benchmark: bases/s of the k-mer counting hot path on MI355X.

Workload (BASELINE.json metric: "bases/sec at k=31, 10M x 150bp"; SURVEY.md 8d "S2"):
  10,000,000 reads x 150 bp sampled (both strands) from a 10 Mbp iid genome (seeds 3/4),
  FASTA text ~1.6 GB, k=31, forward-strand keys (reference behaviour), -c 10, -s 100:
  the reference's Chunker cut points (>= 100 MiB chunks, ~16 of them) are the filter units.
One "step" = the whole sample once: for every chunk  raw FASTA bytes (already resident in HBM)
  -> GPU parse -> 2-bit pack -> count -> keep count >= 10 -> add into the running table,
  then the sorted (key,count) export of the merged table on the device (+ for N > 1 the
  key-range all-to-all merge across ranks over RCCL).  value = bases of all ranks / time.
N > 1 (weak scaling): every rank counts its own 10M reads (same genome, disjoint read
  indices), then the ranks merge their tables.

Extra objects on the JSON line:
  roofline     dominant kernel = the LDS count kernel (mk_sk_count_k); achieved = algorithmic
               bytes per launch (windows * 16 B: one 8-byte key compare + one 8-byte count
               read-modify-write per window, SURVEY.md 8d / DESIGN.md) / mean launch time measured
               with HIP events on the engine's stream; peak = 8 TB/s HBM3E.  `stage_*` repeat the
               calculation for the whole counting stage (partition + count kernels, + the 0.25 B
               per symbol packed read).
  cpu_baseline the CPU oracle (Python restatement of the reference, oracle/cpu_ref.py) timed on
               this box's host cores on a bounded sample at the same coverage.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np
import torch
import torch.distributed as dist

GENOME, GENOME_SEED, READS, READ_LEN, READ_SEED = 10_000_000, 3, 10_000_000, 150, 4
K, MIN_COUNT, CHUNK_MIB = 31, 10, 100
HBM_PEAK_GBS = 8000.0


def cpu_sample_worker(args):
    """One CPU 'chunk': reads at the benchmark's coverage from a small genome, counted by the oracle."""
    from oracle import cpu_ref
    from mercat2_amd import native
    idx, reads, k, c = args
    data = native.synth_reads(reads, 100 + idx, reads, READ_LEN, 200 + idx).tobytes()
    t0 = time.perf_counter()
    table = cpu_ref.count_text(data, k, c)
    return reads * READ_LEN, time.perf_counter() - t0, len(table)


def cpu_baseline(k, c):
    """Bounded CPU run of the oracle: P processes, each one chunk of 40k reads at ~150x coverage
    (the cache-friendliest case for the dict, i.e. generous to the CPU)."""
    import multiprocessing as mp
    cores = max(1, min(os.cpu_count() or 1, 16))
    reads = 40_000
    rounds = 2
    jobs = [(i, reads, k, c) for i in range(cores * rounds)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(cpu_sample_worker, jobs)
    wall = time.perf_counter() - t0
    bases = sum(r[0] for r in res)
    busy = sum(r[1] for r in res)
    return {"value": bases / wall, "unit": "bases/s", "cores": cores, "kind": "port",
            "sample": "%d chunks x %d reads x %d bp (genome %d bp per chunk, ~150x), k=%d, c=%d, pure-Python oracle, "
                      "%d processes; per-core rate %.3g bases/s" % (len(jobs), reads, READ_LEN, reads, k, c, cores,
                                                                    bases / busy)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=READS, help="reads per rank (default: the BASELINE workload)")
    ap.add_argument("--k", type=int, default=K)
    ap.add_argument("--genome", type=int, default=GENOME, help="genome length of the synthetic sample")
    ap.add_argument("--genome-seed", type=int, default=GENOME_SEED)
    ap.add_argument("--read-seed", type=int, default=READ_SEED)
    ap.add_argument("--sub-ppm", type=int, default=0, help="per-base substitution rate, parts per million (S2e: 10000)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--canonical", action="store_true",
                    help="count min(kmer, reverse complement) (BASELINE config 3 names it; an opt-in extension, not the "
                         "reference's forward-strand behaviour -- the default run keeps that)")
    ap.add_argument("--contexts", type=int, default=int(os.environ.get("MK_BENCH_CONTEXTS", "0")),
                    help="engine contexts (HIP streams) per GPU; chunks are dealt round-robin and counted "
                         "concurrently, the tables are merged on the device at the end of the step")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    backend = os.environ.get("MK_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of the N>1 path on a 1-GPU box
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from mercat2_amd import native
    from mercat2_amd.chunker import chunk_offsets
    from mercat2_amd.dist import merge_ranks

    # ---- synthetic input: generate on the host, cut like the reference Chunker, move to HBM
    k = args.k
    t0 = time.perf_counter()
    host = native.synth_reads(args.genome, args.genome_seed, args.reads, READ_LEN, args.read_seed, args.sub_ppm, rank * args.reads)
    offs = chunk_offsets(host, CHUNK_MIB * 1024 * 1024) if host.nbytes >= CHUNK_MIB * 1024 * 1024 else [0, host.nbytes]
    text = torch.from_numpy(host).to(dev)
    gen_s = time.perf_counter() - t0
    del host
    bases_rank = args.reads * READ_LEN
    windows_rank = args.reads * (READ_LEN - k + 1)

    from concurrent.futures import ThreadPoolExecutor
    nctx = args.contexts if args.contexts > 0 else native.default_streams(args.k, native.ALPHABET_NT2)
    ctxs = [native.Counter(k, native.ALPHABET_NT2, device=local, canonical=args.canonical and k <= 32) for _ in range(nctx)]
    ctx = ctxs[0]
    pool = ThreadPoolExecutor(nctx) if nctx > 1 else None
    key_bits = 2 * k
    out_cap = (2 * args.genome + 1024) * (1 if args.sub_ppm == 0 else 12)  # distinct forward-strand k-mers of both strands, upper bound
    out_keys = torch.empty(out_cap, dtype=torch.int64, device=dev)
    out_cnts = torch.empty(out_cap, dtype=torch.int64, device=dev)
    base_ptr = text.data_ptr()
    chunks = list(zip(offs[:-1], offs[1:]))

    def count_share(i):
        c = ctxs[i]
        c.reset()
        for a, b in chunks[i::nctx]:  # every chunk is filtered on its own (the per-chunk -c rule)
            c.count_device(base_ptr + a, b - a, MIN_COUNT)

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

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    for c in ctxs:
        c.reset_stats()
        c.set_profiling(True)
    fence()
    t0 = time.perf_counter()
    rows = 0
    for _ in range(args.steps):
        rows = step()
    fence()
    dt = time.perf_counter() - t0
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
        for a, b in chunks:
            c.count_device(base_ptr + a, b - a, MIN_COUNT)
        torch.cuda.synchronize()
        solo = c.stats()
        c.set_profiling(False)

    red_dev = dev if backend == "nccl" else torch.device("cpu")
    t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    r = torch.tensor([rows], dtype=torch.int64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
    dt = float(t.item())
    total_rows = int(r.item())

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = bases_rank * world * args.steps / dt
        launches = max(1, st["n_count"])
        ms_launch = st["ms_count"] / launches
        packed_windows = st["windows"] - st["exotic_windows"]
        alg_bytes = packed_windows * 16  # over all timed launches of the count kernel
        achieved = alg_bytes / launches / (ms_launch * 1e-3) / 1e9 if ms_launch > 0 else 0.0
        stage_ms = (st["ms_count"] + st["ms_part"]) / launches
        stage_bytes = packed_windows * 16 + st["symbols"] * 0.25
        stage_achieved = stage_bytes / launches / (stage_ms * 1e-3) / 1e9 if stage_ms > 0 else 0.0
        kernel_name = {"hash64": "mk_sk_count_k" if 18 <= k <= 32 else "mk_part_count_k", "dense": "mk_count_dense_k",
                       "byref": "mk_count_byref_k", "ref128": "mk_count_ref128_k"}.get(st["mode_name"], "?")
        if os.environ.get("MK_NO_PARTITION"):
            kernel_name = "mk_count_hash64_k"
        # HBM bytes per launch of that kernel from the committed PMC passes (FETCH_SIZE / WRITE_SIZE in
        # separate rocprofv3 runs of this command, gfx950 correction applied: tools/trim_profiles.py)
        traffic, traffic_src = None, None
        pmc_file = ROOT / "profiles" / "round1_pmc_traffic.json"
        if pmc_file.exists() and args.reads == READS and k == K:
            pmc = json.loads(pmc_file.read_text())
            if kernel_name in pmc:
                traffic = pmc[kernel_name]["hbm_bytes_per_launch"]
                traffic_src = "profiles/round1_pmc_traffic.csv"
        line = {
            "metric": "bases/sec at k=31, 10Mx150bp", "value": value, "unit": "bases/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%s: %d reads x %d bp per GPU from a %d bp genome, k=%d, -c %d, -s %d (%d chunks), "
                                   "%s keys" % ("S2" if (args.reads, args.genome, k, args.sub_ppm) == (READS, GENOME, K, 0) else "custom",
                                                            args.reads, READ_LEN, args.genome, k, MIN_COUNT, CHUNK_MIB, len(offs) - 1,
                                                            "canonical" if (args.canonical and k <= 32) else "forward-strand"),
                       "reads_per_gpu": args.reads, "read_len": READ_LEN, "k": k, "min_count": MIN_COUNT,
                       "chunk_mib": CHUNK_MIB, "chunks": len(offs) - 1, "mode": st["mode_name"], "contexts_per_gpu": nctx,
                       "parallelism": "chunks->ranks, key-range all-to-all merge" if world > 1 else "1 GPU"},
            "distinct_kmers_per_s": total_rows * args.steps / dt,
            "distinct_prefilter_per_s": st["distinct"] * world / dt,  # distinct keys per chunk, before the -c filter
            "rows": total_rows,
            "kernel_ms_per_step": {n: st["ms_" + n] / args.steps for n in ("parse", "pack", "part", "count", "exotic", "filter", "export")},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "launches": launches, "ms_per_launch": ms_launch,
                         "algorithmic_bytes_per_launch": alg_bytes / launches,
                         "stage_ms_per_launch": stage_ms, "stage_achieved": stage_achieved,
                         "stage_frac": stage_achieved / HBM_PEAK_GBS},
            "input_gen_s": gen_s,
        }
        if solo and solo["n_count"] and solo["ms_count"] > 0:
            solo_ms = solo["ms_count"] / solo["n_count"]
            solo_ach = (solo["windows"] - solo["exotic_windows"]) * 16 / solo["n_count"] / (solo_ms * 1e-3) / 1e9
            # (not `achieved`: that one is measured inside the timed region, where two contexts share the GPU)
            line["roofline"]["one_context"] = {"ms_per_launch": solo_ms, "achieved": solo_ach, "frac": solo_ach / HBM_PEAK_GBS,
                                               "note": "same kernel, untimed extra pass with one context"}
        if not args.no_cpu and world == 1:
            line["cpu_baseline"] = cpu_baseline(k, MIN_COUNT)
            line["speedup_vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
