"""Two ranks (processes) sharing the one GPU of the test box: chunks dealt round-robin, every
rank filters its own chunks, then the key-range all-to-all merge (mercat2_amd.dist.merge_ranks).
The concatenation of the ranks' exports in rank order must equal the single-process table.
Backend gloo here (RCCL refuses two ranks on one device); the exchange logic is the same."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import read_input

pytestmark = pytest.mark.gpu
WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _data():
    from mercat2_amd import native
    reads = native.synth_reads(150_000, 31, 120_000, 150, 32).tobytes()
    odd = read_input("edge_lengths.fa")  # N / IUPAC windows -> by-reference rows cross ranks too
    # the all-T 32-mer (the packed table's free-slot mark, kept beside the table) in the FIRST chunk, which
    # rank 0 counts: after the exchange it stands in the middle of the rows the last rank receives
    return b">polyT\n" + b"T" * 200 + b"\n" + reads + odd


def _chunks(data):
    from mercat2_amd.chunker import chunk_offsets
    offs = chunk_offsets(data, 3_000_000)
    return list(zip(offs[:-1], offs[1:]))


def _worker(rank, port, k, c, out):
    import torch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from mercat2_amd import native
        from mercat2_amd.dist import merge_ranks
        data = _data()
        with native.Counter(k, native.ALPHABET_NT2, device=0) as ctx:
            for a, b in _chunks(data)[rank::WORLD]:
                ctx.count_chunk(memoryview(data)[a:b], c)
            merge_ranks(ctx, 2 * k, device=torch.device("cuda", 0))
            kmers, counts = ctx.export()
        out[rank] = (kmers, counts)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,c", [(31, 2), (21, 1), (32, 2)])
def test_two_ranks_equal_one(k, c):
    from mercat2_amd import native
    data = _data()
    with native.Counter(k, native.ALPHABET_NT2) as ctx:
        for a, b in _chunks(data):
            ctx.count_chunk(memoryview(data)[a:b], c)
        want_k, want_c = ctx.export()
    assert len(_chunks(data)) >= 4
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker, args=(port, k, c, out), nprocs=WORLD, join=True)
        parts = [out[r] for r in range(WORLD)]
    # packed rows: rank order == key order; by-reference rows all sit on rank 0, so merge by key
    keys = np.concatenate([p[0].reshape(-1, k).view("S%d" % k).reshape(-1) for p in parts])
    cnts = np.concatenate([p[1] for p in parts])
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(keys[order], want_k.view("S%d" % k).reshape(-1))
    assert np.array_equal(cnts[order], want_c)
    # ownership: every packed (ACGT-only) row of rank 1 is above every packed row of rank 0
    def packed(p):
        ks = p[0].reshape(-1, k)
        ok = np.isin(ks, np.frombuffer(b"ACGT", dtype=np.uint8)).all(axis=1)
        return ks[ok].view("S%d" % k).reshape(-1)
    p0, p1 = packed(parts[0]), packed(parts[1])
    assert p0.size and p1.size and np.sort(p0)[-1] < np.sort(p1)[0]


# ---------------------------------------------------------------- two-word keys travel on the device
def _worker_k63(rank, port, k, c, out):
    import torch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from mercat2_amd import native
        from mercat2_amd import dist as mkdist

        def no_objects(*a, **kw):  # clean reads have no text rows: nothing may be pickled
            raise AssertionError("rows were sent as pickled objects")
        dist.gather_object = dist.all_gather_object = no_objects
        data = native.synth_reads(150_000, 31, 120_000, 150, 32).tobytes()
        with native.Counter(k, native.ALPHABET_NT2, device=0) as ctx:
            assert ctx.words_per_key() == 2
            for a, b in _chunks(data)[rank::WORLD]:
                ctx.count_chunk(memoryview(data)[a:b], c)
            mkdist.merge_ranks(ctx, 2 * k, device=torch.device("cuda", 0))
            kmers, counts = ctx.export()
        out[rank] = (kmers, counts)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,c", [(63, 2), (33, 1)])
def test_two_word_keys_exchange_on_device(k, c):
    """33..64-mers: rows are {hi, lo, count} on the device, re-partitioned by the range of hi with the same
    all_to_all as one-word keys; rank order == key order."""
    from mercat2_amd import native
    data = native.synth_reads(150_000, 31, 120_000, 150, 32).tobytes()
    with native.Counter(k, native.ALPHABET_NT2) as ctx:
        for a, b in _chunks(data):
            ctx.count_chunk(memoryview(data)[a:b], c)
        want_k, want_c = ctx.export()
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker_k63, args=(port, k, c, out), nprocs=WORLD, join=True)
        parts = [out[r] for r in range(WORLD)]
    assert parts[0][1].size and parts[1][1].size
    assert np.array_equal(np.concatenate([p[0] for p in parts]), want_k)  # concatenation in rank order is sorted
    assert np.array_equal(np.concatenate([p[1] for p in parts]), want_c)


# ------------------------------------------------ a one-chunk sample split over the ranks (filter after merge)
def _worker_single(rank, port, k, c, out):
    import torch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from mercat2_amd import native
        from mercat2_amd import dist as mkdist
        data = _data()
        with native.Counter(k, native.ALPHABET_NT2, device=0) as ctx:
            mkdist.count_single_chunk(ctx, data, c, 2 * k, device=torch.device("cuda", 0))
            kmers, counts = ctx.export()
        out[rank] = (kmers, counts)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,c", [(31, 3), (21, 10), (63, 2), (5, 10)])
def test_single_chunk_sample_split_over_ranks(k, c):
    """record ranges, no per-range filter, min_count after the merge == the reference's
    find_kmers on the whole file (lib/mercat2_kmers.py:73-76)."""
    from mercat2_amd import native
    data = _data()
    with native.Counter(k, native.ALPHABET_NT2) as ctx:
        ctx.count_chunk(data, c)  # one chunk, one filter
        want_k, want_c = ctx.export()
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker_single, args=(port, k, c, out), nprocs=WORLD, join=True)
        parts = [out[r] for r in range(WORLD)]
    keys = np.concatenate([p[0].reshape(-1, k).view("S%d" % k).reshape(-1) for p in parts])
    cnts = np.concatenate([p[1] for p in parts])
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(keys[order], want_k.view("S%d" % k).reshape(-1))
    assert np.array_equal(cnts[order], want_c)


def test_record_ranges_cut_at_headers():
    from mercat2_amd.dist import record_ranges
    data = b"pre\nACGT\n>a\nAC\nGT\n>b x>y\nTTTT\n>c\n" + b"A" * 50 + b"\n>d\nC\n"
    for parts in (1, 2, 3, 5, 9):
        rr = record_ranges(data, parts)
        assert rr[0][0] == 0 and rr[-1][1] == len(data) and all(a[1] == b[0] for a, b in zip(rr, rr[1:]))
        for a, _ in rr[1:]:
            assert a == len(data) or (data[a:a + 1] == b">" and data[a - 1:a] == b"\n")


# ------------------------------------------------------------------ bench.py starts and proves its ranks
def test_bench_launches_ranks_and_strong_rows_match(tmp_path):
    """`bench.py --gpus 2` without a launcher starts two ranks itself (gloo rehearsal: both on the one GPU);
    strong scaling (chunks of ONE sample dealt i mod N) ends with the rows of the 1-rank run."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    common = ["--steps", "1", "--warmup", "0", "--no-cpu", "--no-file-leg", "--reads", "2000000", "--genome", "1000000"]
    env = dict(os.environ, MK_BENCH_BACKEND="gloo")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)

    def run(extra):
        p = subprocess.run([sys.executable, str(ROOT / "bench.py")] + common + extra, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, p.stdout
        return json.loads(lines[0])
    one = run(["--gpus", "1"])
    two = run(["--gpus", "2", "--scaling", "strong"])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["ranks_seen"] == 2 and two["scaling"] == "strong"
    assert two["config"]["chunks"] >= 3
    assert two["rows"] == one["rows"] > 0
    assert two["also"]["scaling"] == "weak" and two["also"]["rows"] >= one["rows"]
    # per-rank phases and wire bytes of the timed region, one entry per rank
    assert [r["rank"] for r in two["per_rank"]] == [0, 1]
    assert all(r["count_ms"] > 0 and r["merge_ms"] > 0 and r["wire_bytes_sent"] > 0 for r in two["per_rank"])
    assert sum(r["rows_owned"] for r in two["per_rank"]) == one["rows"]
    # ... and the product's one-process path, timed in a fresh child after the ranks: both transports
    sp = two["also_single_process"]
    for transport in ("peer_copies", "rccl"):
        assert "error" not in sp[transport], sp[transport]
        assert sp[transport]["rows"] == one["rows"] and sp[transport]["n_gpus"] == 2
        assert sp[transport]["merge_devices"]["rccl"] == (1 if transport == "rccl" else 0)


def test_bench_single_process_drives_several_gpus(tmp_path):
    """`bench.py --gpus 2 --single-process`: ONE process, the C ABI's mk_merge_devices (rehearsal: both "GPUs" are
    device 0).  Strong scaling ends with the rows of the 1-GPU run; weak (timed after it) with at least as many."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    common = ["--steps", "1", "--warmup", "0", "--no-cpu", "--no-file-leg", "--no-configs", "--reads", "2000000", "--genome", "1000000"]
    env = dict(os.environ, MK_BENCH_SHARE_DEVICE="1")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)

    def run(extra):
        p = subprocess.run([sys.executable, str(ROOT / "bench.py")] + common + extra, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, p.stdout
        return json.loads(lines[0])
    one = run(["--gpus", "1"])
    two = run(["--gpus", "2", "--single-process", "--scaling", "strong"])
    assert two["n_gpus"] == 2 and two["single_process"] and not two["rccl"] and two["scaling"] == "strong"
    assert two["rows"] == one["rows"] > 0
    assert two["merge_devices"]["contexts"] == 2 and two["merge_devices"]["rows_out"] == one["rows"]
    assert two["also"]["scaling"] == "weak" and two["also"]["rows"] >= one["rows"]


def _worker_balanced(rank, port, k, c, out):
    import torch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from mercat2_amd import native
        from mercat2_amd.dist import merge_ranks
        data = _data()
        with native.Counter(k, native.ALPHABET_NT2, device=0, canonical=True) as ctx:
            for a, b in _chunks(data)[rank::WORLD]:
                ctx.count_chunk(memoryview(data)[a:b], c)
            merge_ranks(ctx, 2 * k, device=torch.device("cuda", 0), balanced=True)
            kmers, counts = ctx.export()
        out[rank] = (kmers, counts)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,c", [(31, 2), (63, 1)])
def test_sampled_splitters_share_canonical_rows_evenly(k, c):
    """Canonical keys crowd the low end of the key space (min(key, revcomp) starts with A or C): with sampled owner
    bounds (merge_ranks(balanced=True)) both ranks end up with about half of the rows, and the table is unchanged."""
    from mercat2_amd import native
    data = _data()
    with native.Counter(k, native.ALPHABET_NT2, canonical=True) as ctx:
        for a, b in _chunks(data):
            ctx.count_chunk(memoryview(data)[a:b], c)
        want_k, want_c = ctx.export()
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker_balanced, args=(port, k, c, out), nprocs=WORLD, join=True)
        parts = [out[r] for r in range(WORLD)]
    keys = np.concatenate([p[0].reshape(-1, k).view("S%d" % k).reshape(-1) for p in parts])
    cnts = np.concatenate([p[1] for p in parts])
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(keys[order], want_k.view("S%d" % k).reshape(-1))
    assert np.array_equal(cnts[order], want_c)
    share = parts[0][1].size / max(1, cnts.size)
    assert 0.40 < share < 0.60, share


# ------------------------------------------------------------------------------- the RCCL backend itself
_RCCL_SCRIPT = r"""
import os, sys
sys.path.insert(0, os.environ["MK_ROOT"])
import numpy as np, torch, torch.distributed as dist
from mercat2_amd import native
from mercat2_amd import dist as mkdist
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)      # backend "nccl" IS RCCL on ROCm
assert dist.get_backend() == "nccl"
ones = torch.ones(1, dtype=torch.int64, device=dev)
dist.all_reduce(ones)
assert int(ones.item()) == 1
data = native.synth_reads(150_000, 31, 60_000, 150, 32).tobytes() + b">polyT\n" + b"T" * 100 + b"\n>odd\nACGTNNACGTRYACGTACGTACGTACGTACGTAACC\n"
for k, c in ((31, 2), (32, 1), (63, 2), (5, 1)):
    with native.Counter(k, native.ALPHABET_NT2, device=0) as ctx:
        ctx.count_chunk(data, c)
        want = ctx.export()
        # the whole exchange -- export on the device, split points, the two all_to_all_single calls, import -- with
        # the one rank sending everything to itself over RCCL
        rows = mkdist.merge_ranks(ctx, 2 * k, device=dev, always=True)
        got = ctx.export()
    assert rows == want[1].size, (k, rows, want[1].size)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), k
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_rccl_backend_runs_the_exchange_on_one_rank(tmp_path):
    """init_process_group("nccl", device_id=...) and the exchange's collectives (all_reduce, all_to_all_single with
    split sizes on int64 (rows, words + 1) tensors) on RCCL itself -- one rank, which is all one GPU allows; the
    2-rank tests above use gloo for the same logic."""
    import subprocess
    import sys
    from conftest import ROOT
    script = tmp_path / "rccl_one.py"
    script.write_text(_RCCL_SCRIPT)
    env = dict(os.environ, MK_ROOT=str(ROOT), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, (p.stdout[-1000:], p.stderr[-3000:])
