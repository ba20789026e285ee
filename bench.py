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


def host_cores():
    """(logical CPUs this process may run on, physical cores among them, CPU quota of the cgroup or None)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    seen = set()
    for cpu in allowed:
        try:
            sib = Path("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % cpu).read_text().strip()
        except OSError:
            sib = str(cpu)
        seen.add(sib)
    quota = None
    try:
        q, per = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    return len(allowed), max(1, len(seen)), quota


def cpu_baseline(k, c, genome, gseed, rseed, chunk_reads, want_cores=0):
    """Bounded CPU run of the oracle on the benchmark's own reads: the first chunk of the sample
    (chunk_reads reads) is split into slices, one per process at a time; every process counts its slice
    as find_kmers would (dict of strings, per-file filter).  The slices' dicts (~4.5 M keys each at
    S2) are far out of cache, as the real chunk's 20 M keys are; smaller dicts are, if anything, kind
    to the CPU.  P = the physical cores this process may use, capped by the
    cgroup's CPU quota when there is one; --cpu-cores overrides."""
    import multiprocessing as mp
    logical, physical, quota = host_cores()
    cores = physical if quota is None else max(1, min(physical, int(quota + 0.5)))
    if want_cores > 0:
        cores = want_cores
    per = max(1000, chunk_reads // 16)  # the same slice size whatever the core count (~0.5 GB of dict per process at S2)
    rounds = 3                          # slices per process, one after the other: ~10-30 s of CPU work per core
    jobs = [(genome, gseed, rseed, i * per, per, k, c) for i in range(cores * rounds)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(cpu_sample_worker, jobs)
    wall = time.perf_counter() - t0
    bases = sum(r[0] for r in res)
    busy = sum(r[1] for r in res)
    out = {"value": bases / wall, "unit": "bases/s", "cores": cores, "cores_available": logical, "physical_cores": physical,
           "cpu_quota": quota, "kind": "port",
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


def launch_ranks(n, argv, timeout_s):
    """Start n copies of this script, one per GPU, and relay rank 0's line.  This (parent) process
    never touches the GPU and execs nothing: the children are fresh processes.  A watchdog ends the run
    when the ranks have not finished after timeout_s seconds (a rank stuck in RCCL init or in a collective
    would otherwise hang until the driver kills the job): the children are killed and the last lines each
    rank wrote to stderr are shown."""
    import tempfile
    import threading
    port = free_port()
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MK_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        log = tempfile.TemporaryFile(mode="w+b")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=log))
    timed_out = threading.Event()

    def watchdog():
        deadline = time.monotonic() + timeout_s
        while time.monotonic() < deadline:
            if all(p.poll() is not None for p in procs):
                return
            time.sleep(0.5)
        timed_out.set()
        for p in procs:  # exactly the children started above, by handle
            if p.poll() is None:
                p.kill()

    wd = threading.Thread(target=watchdog, daemon=True)
    wd.start()
    out, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    wd.join()

    def tail(log, lines=15):
        log.seek(0)
        return b"\n".join(log.read().splitlines()[-lines:]).decode(errors="replace")
    # exactly one line on stdout: rank 0's JSON (whatever else a library printed there goes to stderr)
    for ln in out.decode().splitlines():
        (sys.stdout if ln.startswith("{") else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc]
    if timed_out.is_set() or bad:
        for r, log in enumerate(logs):
            sys.stderr.write("---- rank %d (exit %s), last lines of stderr:\n%s\n" % (r, rcs[r], tail(log)))
        if timed_out.is_set():
            raise SystemExit("bench.py: the ranks did not finish within --rank-timeout %d s; killed" % timeout_s)
        raise SystemExit("bench.py: rank(s) failed: %s" % bad)
    for r, log in enumerate(logs):  # (warnings of a good run still reach the terminal)
        text = tail(log, 5)
        if text.strip():
            sys.stderr.write("[rank %d] %s\n" % (r, text))


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
    ap.add_argument("--rank-timeout", type=int, default=900,
                    help="self-launched ranks (--gpus N without a launcher): seconds after which they are killed and their stderr shown")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N driven by THIS one process through the C ABI (mk_merge_devices: peer copies over xGMI) "
                         "instead of one process per GPU over RCCL")
    ap.add_argument("--merge-rccl", action="store_true",
                    help="--single-process: the segments of mk_merge_devices travel over RCCL (MK_MERGE_RCCL) instead of peer copies")
    ap.add_argument("--no-share", action="store_true",
                    help="every context sums its chunks' survivors into a table of its own (as up to round 3) instead of the "
                         "contexts of a GPU sharing the first one's (mk_share_table)")
    ap.add_argument("--no-single-leg", action="store_true",
                    help="N > 1, one process per GPU: do not run the one-process product path (mk_merge_devices) in a fresh child afterwards")
    ap.add_argument("--cpu-cores", type=int, default=0, help="processes of the cpu_baseline leg (default: the physical cores available)")
    ap.add_argument("--no-configs", action="store_true", help="skip the short runs of the other BASELINE configs after the main region")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-file-leg", action="store_true", help="skip the file-to-TSV leg")
    ap.add_argument("--no-also", action="store_true", help="N > 1: do not time the other scaling mode after the main region")
    ap.add_argument("--canonical", action="store_true",
                    help="count min(kmer, reverse complement) (BASELINE config 3 names it; an opt-in extension, not the "
                         "reference's forward-strand behaviour -- the default run keeps that)")
    ap.add_argument("--contexts", type=int, default=int(os.environ.get("MK_BENCH_CONTEXTS", "0")),
                    help="engine contexts (HIP streams) per GPU; chunks are dealt round-robin and counted "
                         "concurrently, the tables are merged on the device at the end of the step")
    args = ap.parse_args(sys.argv[1:] + os.environ.get("MK_BENCH_EXTRA", "").split())  # (MK_BENCH_EXTRA: more flags, for A/B scripts)

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    single = bool(args.single_process) and args.gpus > 1
    if "RANK" not in os.environ and args.gpus > 1 and not single:
        # no launcher: be one (before torch.cuda / HIP is touched in this process)
        return launch_ranks(args.gpus, sys.argv[1:], args.rank_timeout)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MK_BENCH_TEST_HANG") == str(rank) and os.environ.get("MK_BENCH_CHILD"):
        sys.stderr.write("rank %d: (test) pretending to hang before touching the GPU\n" % rank)
        sys.stderr.flush()
        time.sleep(3600)  # tests/test_bench_launcher.py: the launcher's watchdog must end this
    if single and world != 1:
        raise SystemExit("--single-process is one process for all GPUs: do not start it under a launcher")
    if not single and world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    backend = os.environ.get("MK_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of the N>1 path on a 1-GPU box
    ndev = torch.cuda.device_count()
    if single:
        # MK_BENCH_SHARE_DEVICE=1: rehearsal on a one-GPU box, every "GPU" is device 0 (contexts side by side)
        share = bool(os.environ.get("MK_BENCH_SHARE_DEVICE"))
        if args.gpus > ndev and not share:
            raise SystemExit("%d GPUs asked for, %d visible (MK_BENCH_SHARE_DEVICE=1 rehearses on fewer)" % (args.gpus, ndev))
        devices = [i % ndev for i in range(args.gpus)] if share else list(range(args.gpus))
    else:
        if backend == "nccl" and world > ndev:
            raise SystemExit("%d ranks but %d GPU(s): RCCL needs one device per rank (MK_BENCH_BACKEND=gloo rehearses on fewer)" % (world, ndev))
        devices = [local % ndev]
    local = devices[0]
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
    ngpu = len(devices) * world  # GPUs of the whole job
    part = rank if world > 1 else None  # which share of the job this process has (one GPU per rank, or all of them)

    class Engine:
        """The contexts of this process for one (k, canonical): nctx per device it drives."""

        def __init__(self, k, canonical, genome, sub_ppm):
            self.k, self.canonical = k, canonical
            self.by_dev = [[native.Counter(k, native.ALPHABET_NT2, device=d, canonical=canonical) for _ in range(nctx)] for d in devices]
            self.leaders = [c[0] for c in self.by_dev]
            self.all = [c for cs in self.by_dev for c in cs]
            # the contexts of one GPU upsert their chunks' survivors into ONE running table, the leader's (mk_share_table)
            self.shared = not args.no_share and self.leaders[0].stats()["mode_name"] == "hash64"
            if self.shared:
                for cs in self.by_dev:
                    for c in cs[1:]:
                        c.share_table(cs[0])
            self.pool = ThreadPoolExecutor(len(self.all)) if len(self.all) > 1 else None
            self.words = self.leaders[0].words_per_key()
            self.key_bits = 2 * k
            cap = (2 * genome + 1024) * (1 if sub_ppm == 0 else 12)  # distinct forward-strand k-mers of both strands, upper bound
            self.out_cap = cap
            self.out = [(torch.empty(cap * self.words, dtype=torch.int64, device="cuda:%d" % d),
                         torch.empty(cap, dtype=torch.int64, device="cuda:%d" % d)) for d in devices]
            self.merge_stats = None
            self.phase = {"count_s": 0.0, "merge_s": 0.0, "export_s": 0.0, "steps": 0}  # this process, summed over steps

        def close(self):
            if self.pool:
                self.pool.shutdown()
                self.pool = None
            for c in self.all:
                c.close()
            self.all, self.leaders, self.by_dev = [], [], []

        def make_step(self, parts_by_dev):
            jobs = []  # (context, its chunks)
            for di, cs in enumerate(self.by_dev):
                ptrs = [(p.data_ptr(), p.numel()) for p in parts_by_dev[di]]
                for i, c in enumerate(cs):
                    jobs.append((c, ptrs[i::nctx]))

            def count_share(job):
                c, mine = job
                if not self.shared:
                    c.reset()
                for ptr, n in mine:  # every chunk is filtered on its own (the per-chunk -c rule)
                    c.count_device(ptr, n, MIN_COUNT)

            def reset_all():
                # (a shared table is cleared before any context counts into it again: the owner's reset and the sharers'
                # chunks are not ordered otherwise)
                for c in self.all:
                    c.reset()

            def finish_device(di):
                lead = self.leaders[di]
                for c in self.by_dev[di][1:]:  # sum the other contexts' survivors into the device's first, on the device
                    lead.merge_from(c)

            def export_device(di):
                return self.leaders[di].export_pairs_device(self.out[di][0].data_ptr(), self.out[di][1].data_ptr(), self.out_cap)

            merge_flags = native.MERGE_RANGES | native.MERGE_BALANCED | (native.MERGE_RCCL if args.merge_rccl else 0)

            def step():
                # (every phase ends with a host wait of its own -- the last chunk's read-back, the merge's import, the
                # export's row count -- so plain clocks split the step without adding a synchronisation to it)
                t0 = time.perf_counter()
                if self.shared:
                    reset_all()
                if self.pool is None:
                    count_share(jobs[0])
                else:
                    list(self.pool.map(count_share, jobs))
                if len(devices) == 1:
                    finish_device(0)
                else:
                    list(self.pool.map(finish_device, range(len(devices))))
                t1 = time.perf_counter()
                if len(devices) > 1:
                    # one process, several GPUs: key-range ownership, peer copies (or RCCL), import at the owners (mk_multi.hip)
                    self.merge_stats = native.merge_devices(self.leaders, merge_flags)
                if world > 1:
                    merge_ranks(self.leaders[0], self.key_bits, device=dev)
                t2 = time.perf_counter()
                if len(devices) == 1:
                    rows = export_device(0)
                else:
                    rows = sum(self.pool.map(export_device, range(len(devices))))  # every owner sorts its own range
                t3 = time.perf_counter()
                ph = self.phase
                ph["count_s"] += t1 - t0
                ph["merge_s"] += t2 - t1
                ph["export_s"] += t3 - t2
                ph["steps"] += 1
                return rows
            return step

    # ---- synthetic input: generate on the host, cut like the reference Chunker, move to HBM
    def load_sample(mode, reads, genome, gseed, rseed, sub_ppm):
        """(device chunk tensors per device of this process, chunks of a sample, bases one step counts over the whole job, gen seconds)."""
        t0 = time.perf_counter()
        parts_by_dev, nchunks = [], 0
        for di, d in enumerate(devices):
            g = (rank if world > 1 else di)  # index of this GPU in the job
            if mode == "weak" or di == 0:
                first = g * reads if mode == "weak" else 0
                host = native.synth_reads(genome, gseed, reads, READ_LEN, rseed, sub_ppm, first)
                offs = chunk_offsets(host, CHUNK_MIB * 1024 * 1024) if host.nbytes >= CHUNK_MIB * 1024 * 1024 else [0, host.nbytes]
                spans = list(zip(offs[:-1], offs[1:]))
                nchunks = len(spans)
            if mode == "weak":
                whole = torch.from_numpy(host).to("cuda:%d" % d)
                parts_by_dev.append([whole[a:b] for a, b in spans])
            else:
                parts_by_dev.append([torch.from_numpy(host[a:b]).to("cuda:%d" % d) for a, b in spans[g::ngpu]])
        for d in devices:
            torch.cuda.synchronize(d)
        total = reads * READ_LEN * (ngpu if mode == "weak" else 1)
        return parts_by_dev, nchunks, total, time.perf_counter() - t0

    def fence():
        if world > 1:
            dist.barrier()
        for d in devices:
            torch.cuda.synchronize(d)

    def timed(eng, step, steps, warmup, profile):
        for _ in range(warmup):
            step()
        if profile:
            for c in eng.all:
                c.reset_stats()
                c.set_profiling(True)
        eng.phase.update(count_s=0.0, merge_s=0.0, export_s=0.0, steps=0)
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

    def sum_stats(ctx_list):
        stats = [c.stats() for c in ctx_list]
        st = dict(stats[0])
        for other in stats[1:]:
            for key, val in other.items():
                if key.startswith(("ms_", "n_")) or key in ("windows", "exotic_windows", "symbols", "raw_bytes", "chunks", "records", "distinct",
                                                              "part_retries", "part_reused", "fused_chunks", "fuse_spilled"):
                    st[key] += val
        return st

    # the reference's tables of the BASELINE workloads (tests/golden/expected_s2.json, made by tests/golden/make_s2_golden.py
    # from the reference's own Chunker + find_kmers): rows, sum of counts, sha256 of the concatenated keys and of the counts
    golden = {}
    try:
        gj = json.loads((ROOT / "tests" / "golden" / "expected_s2.json").read_text())
        for name, wl in (("S2|k31|c10|s100", (READS, GENOME, K, GENOME_SEED, READ_SEED)),
                         ("S1|k21|c10|s100", (1_000_000, 1_000_000, 21, 1, 2)),
                         ("S3|k63|c10|s100", (50_000_000, 50_000_000, 63, 6, 7))):
            if name in gj:
                for canon, tab in ((False, "forward"), (True, "canonical")):
                    if tab in gj[name]:
                        golden[wl + (canon,)] = gj[name][tab]
    except (OSError, KeyError, ValueError):
        pass

    def golden_for(mode, reads, genome, kk, sub_ppm, canon, gseed, rseed):
        if sub_ppm or not (ngpu == 1 or mode == "strong"):
            return None
        return golden.get((reads, genome, kk, gseed, rseed, bool(canon)))

    def verify_rows(rows, mode, reads, genome, kk, sub_ppm, canon, gseed, rseed):
        """True/False when this run is a sample whose table the reference produced; None when it is another workload."""
        g = golden_for(mode, reads, genome, kk, sub_ppm, canon, gseed, rseed)
        return None if g is None else rows == g["rows"]

    def verify_table(eng_, mode, reads, genome, kk, sub_ppm, canon, gseed, rseed):
        """After a timed region: the table the last step left (one GPU: in the leader context), exported once more
        through mk_export and compared with the reference's -- rows, sum of counts, sha256 of keys and counts.
        {"rows": bool, "sum": bool, "digest": bool} or None when the reference holds no table for this workload."""
        import hashlib
        g = golden_for(mode, reads, genome, kk, sub_ppm, canon, gseed, rseed)
        if g is None or ngpu != 1:
            return None
        kmers, counts = eng_.leaders[0].export()
        res = {"rows": int(counts.shape[0]) == g["rows"], "sum": int(counts.sum()) == g["sum"]}
        res["digest"] = (hashlib.sha256(kmers.tobytes()).hexdigest() == g["keys_sha256"] and
                         hashlib.sha256(counts.astype("<u8").tobytes()).hexdigest() == g["counts_sha256"])
        return res

    mode = args.scaling if ngpu > 1 else "weak"
    eng = Engine(k, canonical, args.genome, args.sub_ppm)
    parts, nchunks, total_bases, gen_s = load_sample(mode, args.reads, args.genome, args.genome_seed, args.read_seed, args.sub_ppm)
    step = eng.make_step(parts)
    dt, total_rows, ranks_seen = timed(eng, step, args.steps, args.warmup, True)
    st = sum_stats(eng.all)
    for c in eng.all:
        c.set_profiling(False)
    main_merge_stats = eng.merge_stats
    # per-rank phases of the timed region (DESIGN section 6 holds the model they are to be read against)
    from mercat2_amd import dist as mkdist
    n_ph = max(1, eng.phase["steps"])
    mine = {"rank": rank, "device": local, "count_ms": eng.phase["count_s"] / n_ph * 1e3, "merge_ms": eng.phase["merge_s"] / n_ph * 1e3,
            "export_ms": eng.phase["export_s"] / n_ph * 1e3, "rows_owned": None, "wire_bytes_sent": 0, "wire_bytes_received": 0}
    if world > 1 and mkdist.LAST_MERGE:
        lm = mkdist.LAST_MERGE
        mine.update(rows_owned=lm["rows_owned"], wire_bytes_sent=lm["wire_bytes_sent"], wire_bytes_received=lm["wire_bytes_received"],
                    merge_bucket_ms=lm["bucket_s"] * 1e3, merge_collectives_ms=lm["collectives_s"] * 1e3, merge_import_ms=lm["import_s"] * 1e3)
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        with torch.cuda.device(dev):
            dist.all_gather_object(per_rank, mine)
    verified = verify_rows(total_rows, mode, args.reads, args.genome, k, args.sub_ppm, canonical, args.genome_seed, args.read_seed)
    verified_table = verify_table(eng, mode, args.reads, args.genome, k, args.sub_ppm, canonical, args.genome_seed, args.read_seed) if rank == 0 else None
    if verified_table is not None and not all(verified_table.values()):
        verified = False

    # After the timed region: the same kernels alone on the GPU (one context, one pass), so that the
    # dominant kernel's duration can also be read without the other stream's kernels on its CUs.
    solo = None
    if nctx > 1 and rank == 0:
        c = eng.all[0]
        c.reset()
        c.reset_stats()
        c.set_profiling(True)
        for p in parts[0]:
            c.count_device(p.data_ptr(), p.numel(), MIN_COUNT)
        torch.cuda.synchronize(devices[0])
        solo = c.stats()
        c.set_profiling(False)

    # the other scaling mode, timed after the main region (N > 1 only)
    also = None
    if ngpu > 1 and not args.no_also:
        other_mode = "strong" if mode == "weak" else "weak"
        del parts
        parts2, nchunks2, total2, _ = load_sample(other_mode, args.reads, args.genome, args.genome_seed, args.read_seed, args.sub_ppm)
        dt2, rows2, _ = timed(eng, eng.make_step(parts2), args.steps, 1, False)
        v2 = verify_rows(rows2, other_mode, args.reads, args.genome, k, args.sub_ppm, canonical, args.genome_seed, args.read_seed)
        also = {"scaling": other_mode, "value": total2 * args.steps / dt2, "unit": "bases/s", "ms_per_step": dt2 / args.steps * 1e3,
                "rows": rows2, "verified_rows": v2, "chunks": nchunks2, "note": "timed after the main region, same steps, 1 warm-up"}
        if v2 is False:
            verified = False
        del parts2
        parts = None

    # the other BASELINE configs, each a short run after the main region (one GPU, default workload only):
    # config 3 as worded (canonical keys), config 2 (S1: 1 M reads, k=21) and config 5 (S3: 50 M reads, k=63)
    configs = None
    default_run = (args.reads, args.genome, k, args.sub_ppm, canonical) == (READS, GENOME, K, 0, False)
    if ngpu == 1 and rank == 0 and default_run and not args.no_configs:
        configs = {}
        # (the main engine's contexts go first: HIP spreads a process's streams over a few hardware queues, and with the
        # main engine's two streams still alive the two streams of a leg's engine landed on ONE queue -- their kernels ran
        # one after the other, config 3 measured 14.6-15.5 ms as a leg against 12.7-13.0 ms as a run of its own)
        eng.close()

        def short_run(name, kk, canon, reads, genome, gseed, rseed, steps, parts_in=None, warm=2):
            e2 = Engine(kk, canon, genome, 0)
            try:
                p2, nch, tot, _ = (parts_in, nchunks, total_bases, 0) if parts_in is not None else load_sample("weak", reads, genome, gseed, rseed, 0)
                d2, rows2, _ = timed(e2, e2.make_step(p2), steps, warm, True)  # (warm-up steps: the running tables grow to their size in the first ones)
                s2 = sum_stats(e2.all)
                two = s2["mode_name"] == "hash128"
                bpw = 24 if two else 16
                ms_l = s2["ms_count"] / max(1, s2["n_count"])
                ach = (s2["windows"] - s2["exotic_windows"]) * bpw / max(1, s2["n_count"]) / (ms_l * 1e-3) / 1e9 if ms_l > 0 else 0.0
                configs[name] = {"value": tot * steps / d2, "unit": "bases/s", "ms_per_step": d2 / steps * 1e3, "steps": steps, "warmup": warm,
                                 "rows": rows2, "chunks": nch, "k": kk, "reads": reads, "genome": genome, "canonical": canon,
                                 "mode": s2["mode_name"], "count_kernel_ms_per_launch": ms_l, "count_kernel_frac": ach / HBM_PEAK_GBS,
                                 "bytes_per_window": bpw, "part_retries": s2["part_retries"], "part_reused": s2["part_reused"],
                                 "verified_rows": verify_rows(rows2, "weak", reads, genome, kk, 0, canon, gseed, rseed)}
                vt = verify_table(e2, "weak", reads, genome, kk, 0, canon, gseed, rseed)
                if vt is not None:
                    configs[name]["verified_sum"], configs[name]["verified_digest"] = vt["sum"], vt["digest"]
                    if not all(vt.values()):
                        configs[name]["verified_rows"] = False
                del p2
            finally:
                e2.close()

        short_run("config3_canonical", K, True, READS, GENOME, GENOME_SEED, READ_SEED, max(4, args.steps // 2), parts_in=parts, warm=3)
        if configs["config3_canonical"]["verified_rows"] is False:
            verified = False
        del parts
        parts = None
        torch.cuda.empty_cache()
        short_run("config2_s1_k21", 21, False, 1_000_000, 1_000_000, 1, 2, max(2, args.steps))
        short_run("config5_s3_k63", 63, False, 50_000_000, 50_000_000, 6, 7, 3)
        if any(c_["verified_rows"] is False for c_ in configs.values()):
            verified = False

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
        kernel_name = {"hash64": "mk_sk_count_k" if 12 <= k <= 32 else "mk_part_count_k", "dense": "mk_count_dense_k",
                       "byref": "mk_count_byref_k", "hash128": "mk_sk2_count_k"}.get(st["mode_name"], "?")
        if os.environ.get("MK_NO_PARTITION"):
            kernel_name = "mk_count_hash64_k"
        # HBM bytes per launch of that kernel from the committed PMC passes (FETCH_SIZE / WRITE_SIZE in
        # separate rocprofv3 runs of this command, gfx950 correction applied: tools/trim_profiles.py)
        traffic, traffic_src = None, None
        for name in ("round4_pmc_traffic.json", "round3_pmc_traffic.json", "round2_pmc_traffic.json", "round1_pmc_traffic.json"):
            pmc_file = ROOT / "profiles" / name
            if pmc_file.exists() and args.reads == READS and k == K and not canonical:
                pmc = json.loads(pmc_file.read_text())
                if kernel_name in pmc:
                    traffic = pmc[kernel_name]["hbm_bytes_per_launch"]
                    traffic_src = "profiles/" + name.replace(".json", ".csv")
                    break
        is_s2 = (args.reads, args.genome, k, args.sub_ppm) == (READS, GENOME, K, 0)
        if single:
            par = "one process, %d GPUs: chunks->GPUs, key-range merge by peer copies (mk_merge_devices)" % ngpu
        elif world > 1:
            par = "chunks->ranks, key-range all-to-all merge"
        else:
            par = "1 GPU"
        line = {
            "metric": "bases/sec at k=31, 10Mx150bp", "value": value, "unit": "bases/s", "n_gpus": ngpu,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": mode, "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "ranks_seen": ranks_seen, "rccl": bool(world > 1 and backend == "nccl"), "backend": backend if world > 1 else None,
            "single_process": single,
            "config": {"workload": "%s: %d reads x %d bp %s from a %d bp genome, k=%d, -c %d, -s %d (%d chunks per sample), %s keys"
                                   % ("S2" if is_s2 else "custom", args.reads, READ_LEN,
                                      "per GPU" if mode == "weak" else "in all (one sample, chunks dealt i mod N)", args.genome, k,
                                      MIN_COUNT, CHUNK_MIB, nchunks, "canonical" if canonical else "forward-strand"),
                       "reads_per_gpu": args.reads if mode == "weak" else args.reads / ngpu, "read_len": READ_LEN, "k": k,
                       "min_count": MIN_COUNT, "chunk_mib": CHUNK_MIB, "chunks": nchunks, "mode": st["mode_name"],
                       "contexts_per_gpu": nctx, "shared_table": eng.shared, "parallelism": par},
            "distinct_kmers_per_s": total_rows * args.steps / dt,
            "distinct_prefilter_per_s": st["distinct"] * world / dt,  # distinct keys per chunk, before the -c filter
            "rows": total_rows,
            # the rows the REFERENCE gets for this sample (tests/golden/expected_s2.json); null when the run is not that sample
            "verified_rows": verified,
            # the table of the last timed step exported once more and compared with the reference's sum and sha256 digests
            "verified_sum": None if verified_table is None else verified_table["sum"],
            "verified_digest": None if verified_table is None else verified_table["digest"],
            "kernel_ms_per_step": {n: st["ms_" + n] / args.steps for n in ("parse", "pack", "part", "count", "exotic", "filter", "export")},
            # chunks whose count kernel merged its survivors into the running table itself (no import kernel), and what those set aside
            "fused_chunks_per_step": st["fused_chunks"] / args.steps, "fuse_spilled": st["fuse_spilled"],
            # chunks partitioned a second time (sampled bucket regions too small) / chunks that inherited the regions of the chunk before
            "part_retries": st["part_retries"], "part_reused": st["part_reused"],
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "launches": launches, "ms_per_launch": ms_launch,
                         "algorithmic_bytes_per_launch": alg_bytes / launches, "bytes_per_window": bytes_per_window,
                         "stage_ms_per_launch": stage_ms, "stage_achieved": stage_achieved,
                         "stage_frac": stage_achieved / HBM_PEAK_GBS},
            "input_gen_s": gen_s,
        }
        if main_merge_stats:
            line["merge_devices"] = main_merge_stats  # (of the last timed step)
        if also:
            line["also"] = also
        if configs:
            line["configs"] = configs
        if solo and solo["n_count"] and solo["ms_count"] > 0:
            solo_ms = solo["ms_count"] / solo["n_count"]
            solo_ach = (solo["windows"] - solo["exotic_windows"]) * bytes_per_window / solo["n_count"] / (solo_ms * 1e-3) / 1e9
            # (not `achieved`: that one is measured inside the timed region, where two contexts share the GPU)
            line["roofline"]["one_context"] = {"ms_per_launch": solo_ms, "achieved": solo_ach, "frac": solo_ach / HBM_PEAK_GBS,
                                               "note": "same kernel, untimed extra pass with one context"}
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
        except Exception:  # (a probe, not the product: never fail the bench line over it)
            line["roofline"]["measured_stream_copy"] = None
        if ngpu == 1 and not args.no_file_leg:
            line["file_to_tsv"] = file_to_tsv_leg(args, k, canonical)
        if not args.no_cpu and ngpu == 1:
            chunk_reads = max(1, args.reads // max(1, nchunks))
            line["cpu_baseline"] = cpu_baseline(k, MIN_COUNT, args.genome, args.genome_seed, args.read_seed, chunk_reads, args.cpu_cores)
            line["speedup_vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
            if "file_to_tsv" in line:  # the target is stated on the file-to-TSV window
                cpu = line["cpu_baseline"]["value"]
                line["file_to_tsv"]["plain_vs_cpu_baseline"] = line["file_to_tsv"]["plain_bases_per_s"] / cpu
                line["file_to_tsv"]["gz_vs_cpu_baseline"] = line["file_to_tsv"]["gz_bases_per_s"] / cpu
                line["file_to_tsv"]["vs_cpu_note"] = ("the CPU figure is counting only (text already in memory): against a "
                                                      "CPU run that also had to inflate the .gz the ratio would be larger")
        line["per_rank"] = per_rank
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world > 1 and not args.no_single_leg:
            # every rank has closed its contexts: the GPUs are idle.  The product's own N-GPU path -- ONE process, chunks
            # dealt to the GPUs through the C ABI, mk_merge_devices -- timed in a fresh child process of this one (never
            # an exec of this process), its line folded in here
            line["also_single_process"] = single_process_leg(args, backend)
        print(json.dumps(line))
        sys.stdout.flush()
    if verified is False:
        raise SystemExit("bench.py: a table of a timed workload differs from the reference's table of the same sample "
                         "(rows %d; verified_table %s; configs %s): the result is WRONG"
                         % (total_rows, verified_table, {n_: c_.get("verified_rows") for n_, c_ in (configs or {}).items()}))


def single_process_leg(args, backend):
    """`bench.py --gpus N --single-process` in a child process; the figures of its line that say how the one-process path
    did (value, step, phases of mk_merge_devices, peer access), or {"error": ...}: never fails the parent's line."""
    cmd = [sys.executable, str(Path(__file__).resolve()), "--gpus", str(args.gpus), "--single-process", "--steps", str(args.steps),
           "--warmup", "1", "--scaling", args.scaling, "--reads", str(args.reads), "--k", str(args.k), "--genome", str(args.genome),
           "--genome-seed", str(args.genome_seed), "--read-seed", str(args.read_seed), "--sub-ppm", str(args.sub_ppm),
           "--no-also", "--no-configs", "--no-cpu", "--no-file-leg"]
    if args.canonical:
        cmd.append("--canonical")
    if args.contexts > 0:
        cmd += ["--contexts", str(args.contexts)]
    env = {k_: v_ for k_, v_ in os.environ.items()
           if k_ not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MK_BENCH_CHILD", "GROUP_RANK",
                         "ROLE_RANK", "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    if backend != "nccl":
        env["MK_BENCH_SHARE_DEVICE"] = "1"  # (a rehearsal on fewer GPUs than ranks stays one)
    out = {}
    for label, extra in (("peer_copies", []), ("rccl", ["--merge-rccl"])):
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=min(args.rank_timeout, 300))
            lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not lines:
                out[label] = {"error": "exit %d: %s" % (p.returncode, p.stderr.decode(errors="replace")[-400:])}
                continue
            j = json.loads(lines[-1])
            out[label] = {key: j.get(key) for key in ("value", "unit", "ms_per_step", "n_gpus", "scaling", "rows", "verified_rows",
                                                      "merge_devices", "per_rank", "steps", "warmup")}
            out[label]["wall_s"] = time.perf_counter() - t0
        except subprocess.TimeoutExpired:
            out[label] = {"error": "no line within %d s (the child was killed)" % min(args.rank_timeout, 300)}
        except (OSError, ValueError) as e:
            out[label] = {"error": repr(e)[:300]}
    out["note"] = "one process drives all GPUs through the C ABI (mk_count_device per chunk, mk_merge_devices, mk_export per " \
                  "owner); run in a fresh child after the ranks had closed their contexts; 1 warm-up step"
    return out


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
            best, st_best, tm_best, rows = None, None, None, 0
            for _ in range(2):
                st, tm = {}, {}
                out = os.path.join(d, "S2_counts.tsv")
                lines = []
                t0 = time.perf_counter()
                harness.run_sample("S2", path, out, k, MIN_COUNT, CHUNK_MIB, canonical=canonical, stats=st, report=lines.append, timings=tm)
                dt = time.perf_counter() - t0
                if best is None or dt < best:
                    best, st_best, tm_best = dt, st, tm
                rows = int(lines[0].split(":")[1]) if lines and ":" in lines[0] else 0
            res[name + "_s"] = best
            res[name + "_bases_per_s"] = args.reads * READ_LEN / best
            res[name + "_threads"] = st_best.get("threads")
            res[name + "_file_bytes"] = os.path.getsize(path)
            # where the window went (seconds; mk_file_stats_t + mk_export_stats_t): the parts of the DISPATCHING thread of
            # mk_count_file add up to count_file; count_file + tsv + other = the window
            ex = tm_best.get("export", {})
            parts = {"count_file": st_best.get("s_total"),
                     "count_file_parts": {n_[2:]: st_best.get(n_) for n_ in ("s_setup", "s_wait_io", "s_wait_gpu", "s_scan", "s_feed",
                                                                             "s_retire", "s_drain", "s_merge")},
                     "tsv": tm_best.get("tsv_s"),
                     "tsv_parts": {n_[2:]: ex.get(n_) for n_ in ("s_sort", "s_d2h", "s_format", "s_write")},
                     "tsv_bytes": ex.get("bytes")}
            parts["other"] = best - (parts["count_file"] or 0.0) - (parts["tsv"] or 0.0)  # contexts taken / given back, Python
            res[name + "_breakdown_s"] = parts
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
