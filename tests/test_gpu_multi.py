"""One process, several GPUs, through the C ABI: mk_merge_devices, mk_*_multi, mk_bucket_rows_device /
mk_import_rows_device and mk_count_file over contexts "on different devices".

The test box has ONE GPU: the N-device code path runs with N contexts on device 0 (device list [0, 0, ..]; for
mk_count_file MK_DEVICE_PER_CONTEXT=1 makes every context count as a GPU of its own) -- the same grouping by owner,
copies (device-to-device instead of peer), imports and exports.  Results are compared with the CPU oracle and with
the one-context table (which the rest of the suite pins to the oracle)."""
import os

import numpy as np
import pytest

from conftest import read_input
from mercat2_amd import native
from oracle import cpu_ref

pytestmark = pytest.mark.gpu


def _data():
    reads = native.synth_reads(150_000, 31, 60_000, 150, 32).tobytes()
    odd = read_input("edge_lengths.fa")  # N / IUPAC windows -> rows kept as text
    return b">polyT\n" + b"T" * 200 + b"\n" + reads + odd


def _chunks(data, size=2_000_000):
    from mercat2_amd.chunker import chunk_offsets
    offs = chunk_offsets(data, size)
    return list(zip(offs[:-1], offs[1:]))


def _oracle_chunked(data, k, c):
    total = {}
    for a, b in _chunks(data):
        for key, n in cpu_ref.count_text(data[a:b], k, c).items():
            total[key] = total.get(key, 0) + n
    return total


def _as_dict(kmers, counts, k):
    flat = kmers.tobytes().decode("ascii")
    return dict(zip((flat[i:i + k] for i in range(0, len(flat), k)), counts.tolist()))


@pytest.mark.parametrize("k,c,alphabet", [(31, 2, native.ALPHABET_NT2), (21, 1, native.ALPHABET_NT2), (32, 2, native.ALPHABET_NT2),
                                          (33, 2, native.ALPHABET_NT2), (63, 1, native.ALPHABET_NT2), (3, 10, native.ALPHABET_NT2),
                                          (12, 2, native.ALPHABET_NT2), (70, 2, native.ALPHABET_NT2)])
@pytest.mark.parametrize("n,flags", [(2, native.MERGE_RANGES), (3, native.MERGE_RANGES | native.MERGE_BALANCED), (3, native.MERGE_GATHER),
                                     # the same exchange over RCCL (ncclSend / ncclRecv in one group; the contexts share the one
                                     # device here, so every segment is a send of rank 0 to itself)
                                     (3, native.MERGE_RANGES | native.MERGE_BALANCED | native.MERGE_RCCL), (2, native.MERGE_GATHER | native.MERGE_RCCL)])
def test_merge_devices_equals_the_oracle(k, c, alphabet, n, flags):
    data = _data()
    want = _oracle_chunked(data, k, c)
    ctxs = [native.Counter(k, alphabet, device=0) for _ in range(n)]
    try:
        spans = _chunks(data)
        assert len(spans) >= 4
        for i, (a, b) in enumerate(spans):  # chunk i -> context i mod n, filtered on its own
            ctxs[i % n].count_chunk(memoryview(data)[a:b], c)
        rows_each = [x.rows() for x in ctxs]
        st = native.merge_devices(ctxs, flags)
        assert st["contexts"] == n and st["devices"] == 1
        assert st["rccl"] == (1 if flags & native.MERGE_RCCL else 0)
        kmers, counts = native.export_multi(ctxs)
        assert _as_dict(kmers, counts, k) == want
        assert native.rows_multi(ctxs) == len(want)
        assert st["rows_out"] == len(want)
        if flags & native.MERGE_GATHER:
            assert ctxs[0].rows() == len(want) and all(x.rows() == 0 for x in ctxs[1:])
            assert ctxs[0].to_dict() == want
        else:
            # ownership: the contexts' packed rows are disjoint ascending ranges; rows kept as text sit in context 0
            last = None
            for x in ctxs:
                km, _ = x.export()
                packed = [r.tobytes() for r in km if set(r.tobytes()) <= set(b"ACGT")] if k <= 64 else []
                if x is not ctxs[0]:
                    assert len(packed) == km.shape[0], "text rows outside context 0"
                if packed:
                    assert last is None or last < min(packed)
                    last = max(packed)
            if st["rows_moved"]:
                assert st["bytes_moved"] == st["rows_moved"] * 8 * (ctxs[0].words_per_key() + 1)
        assert sum(rows_each) >= len(want)
    finally:
        for x in ctxs:
            x.close()


@pytest.mark.parametrize("k,c,canonical", [(31, 2, False), (21, 3, False), (31, 2, True), (32, 2, False)])
def test_contexts_of_one_gpu_share_a_running_table(k, c, canonical):
    """mk_share_table: three contexts (streams) of the one GPU count the chunks of a sample side by side, each from a
    host thread of its own, and their count kernels upsert the survivors into ONE table -- the first context's; what a
    context still merges on its own (its first chunk, when it is new) is summed at the end as before.  Two samples one
    after the other through the same contexts (the second finds the tables sized and every chunk fused), then the
    contexts go back to tables of their own.  Equal to the oracle's chunk-by-chunk sum every time."""
    from concurrent.futures import ThreadPoolExecutor
    from mercat2_amd.chunker import chunk_offsets
    from oracle import c_oracle
    n = 3
    ctxs = [native.Counter(k, native.ALPHABET_NT2, device=0, canonical=canonical) for _ in range(n)]
    try:
        for x in ctxs[1:]:
            x.share_table(ctxs[0])
        with pytest.raises(native.MercatHipError):
            ctxs[0].share_table(ctxs[1])  # one level deep
        for seed in (51, 53):
            data = native.synth_reads(80_000, seed, 90_000, 150, seed + 1).tobytes()
            offs = chunk_offsets(data, 1_500_000)
            spans = list(zip(offs[:-1], offs[1:]))
            assert len(spans) >= 8
            parts = [c_oracle.count_dict(data[a:b], k, 0 if canonical else c) for a, b in spans]
            if canonical:
                parts = [{key: m for key, m in cpu_ref.canonical_fold(p).items() if m >= c} for p in parts]
            want = cpu_ref.merge_counts(parts)
            for x in ctxs:  # (the owner is reset before anybody counts into its table again)
                x.reset()

            def share(i):
                for a, b in spans[i::n]:
                    ctxs[i].count_chunk(memoryview(data)[a:b], c)
            with ThreadPoolExecutor(n) as pool:
                list(pool.map(share, range(n)))
            fused = [x.stats()["fused_chunks"] for x in ctxs]
            for x in ctxs[1:]:
                ctxs[0].merge_from(x)
            assert ctxs[0].to_dict() == want, (seed, fused)
            if seed == 53:  # (second sample: hints and table sizes are there, the sharers' rows are in the owner's table)
                assert all(x.rows() < len(want) // 4 for x in ctxs[1:]), [x.rows() for x in ctxs]
        for x in ctxs[1:]:
            x.share_table(None)
        data = native.synth_reads(80_000, 57, 30_000, 150, 58).tobytes()
        for x in ctxs:
            x.reset()
            x.reset_stats()
        ctxs[1].count_chunk(data, c)
        got = c_oracle.count_dict(data, k, 0 if canonical else c)
        if canonical:
            got = {key: m for key, m in cpu_ref.canonical_fold(got).items() if m >= c}
        assert ctxs[1].to_dict() == got and ctxs[0].rows() == 0
    finally:
        for x in ctxs:
            x.close()


def test_balanced_bounds_even_out_a_skewed_table():
    """Keys crowded into the low half of the key space (sequences over A and C only: every key starts with the bits
    00 or 01): equal key ranges leave two of four owners empty, sampled splitters share the rows out."""
    rng = np.random.default_rng(5)
    recs = [">r%d\n%s\n" % (i, "".join("AC"[x] for x in rng.integers(0, 2, 100))) for i in range(20000)]
    data = "".join(recs).encode()
    k, n = 21, 4
    want = cpu_ref.count_text(data, k, 1)
    out = {}
    for name, flags in (("equal", native.MERGE_RANGES), ("balanced", native.MERGE_RANGES | native.MERGE_BALANCED)):
        ctxs = [native.Counter(k, native.ALPHABET_NT2, device=0) for _ in range(n)]
        try:
            quarter = len(recs) // n
            for i in range(n):
                ctxs[i].count_chunk("".join(recs[i * quarter:(i + 1) * quarter]).encode(), 1)
            st = native.merge_devices(ctxs, flags)
            kmers, counts = native.export_multi(ctxs)
            assert _as_dict(kmers, counts, k) == want
            out[name] = st["max_owned"] / max(1, st["rows_out"])
        finally:
            for x in ctxs:
                x.close()
    assert out["equal"] > 0.45, out     # two owners hold everything
    assert out["balanced"] < 0.32, out  # four owners: 0.25 is perfect


@pytest.mark.parametrize("k", [31, 32, 63, 3])
def test_bucket_rows_and_import_rows_round_trip(k):
    """mk_bucket_rows_device -> mk_import_rows_device (what dist.py sends over RCCL): every row once, grouped by owner."""
    import torch
    data = _data()
    with native.Counter(k, native.ALPHABET_NT2, device=0) as a, native.Counter(k, native.ALPHABET_NT2, device=0) as b:
        a.count_chunk(data, 2)
        want_k, want_c = a.export()
        words = a.words_per_key()
        key_bits = 64 if words == 2 else 2 * k
        bounds = native.owner_bounds(key_bits, 5)
        cap = a.rows() + 1
        rows = torch.zeros((cap, words + 1), dtype=torch.int64, device="cuda:0")
        counts = a.bucket_rows_device(bounds, rows.data_ptr(), cap)
        total = sum(counts)
        packed_rows = a.rows() - a.export_exotic()[1].size
        assert total == packed_rows
        # owner of every row: number of bounds <= the first key word (unsigned)
        first = rows[:total, 0].cpu().numpy().view(np.uint64)
        owner = np.searchsorted(np.array(bounds, dtype=np.uint64), first, side="right")
        at = 0
        for j, cnt in enumerate(counts):
            assert np.all(owner[at:at + cnt] == j)
            at += cnt
        b.import_rows_device(rows.data_ptr(), total)
        ek, ec = a.export_exotic()
        b.import_exotic(ek, ec)
        got_k, got_c = b.export()
        assert np.array_equal(got_k, want_k) and np.array_equal(got_c, want_c)
        # size query
        assert a.bucket_rows_device(bounds, 0, 0) == counts


def test_export_multi_refuses_overlapping_contexts():
    data = _data()
    ctxs = [native.Counter(31, native.ALPHABET_NT2, device=0) for _ in range(2)]
    try:
        for x in ctxs:
            x.count_chunk(data, 2)
        with pytest.raises(native.MercatHipError) as e:
            native.export_multi(ctxs)
        assert e.value.code == -4
        native.merge_devices(ctxs)
        kmers, counts = native.export_multi(ctxs)
        with native.Counter(31, native.ALPHABET_NT2, device=0) as one:
            one.count_chunk(data, 2)
            wk, wc = one.export()
        assert np.array_equal(kmers, wk) and np.array_equal(counts, 2 * wc)
    finally:
        for x in ctxs:
            x.close()


def test_write_tsv_multi_is_the_one_context_file(tmp_path):
    data = _data()
    k, c, n = 31, 2, 3
    ctxs = [native.Counter(k, native.ALPHABET_NT2, device=0) for _ in range(n)]
    try:
        with native.Counter(k, native.ALPHABET_NT2, device=0) as one:
            for i, (a, b) in enumerate(_chunks(data)):
                ctxs[i % n].count_chunk(memoryview(data)[a:b], c)
                one.count_chunk(memoryview(data)[a:b], c)
            rows1 = one.write_tsv(tmp_path / "one.tsv", "S")
        native.merge_devices(ctxs, native.MERGE_RANGES | native.MERGE_BALANCED)
        rows = native.write_tsv_multi(ctxs, tmp_path / "multi.tsv", "S")
        assert rows == rows1 > 0
        assert (tmp_path / "multi.tsv").read_bytes() == (tmp_path / "one.tsv").read_bytes()
    finally:
        for x in ctxs:
            x.close()
    # nothing survives: no file
    ctxs = [native.Counter(k, native.ALPHABET_NT2, device=0) for _ in range(2)]
    try:
        ctxs[0].count_chunk(b">a\nACGT\n", 10)
        native.merge_devices(ctxs)
        assert native.write_tsv_multi(ctxs, tmp_path / "none.tsv", "S") == 0
        assert not (tmp_path / "none.tsv").exists()
    finally:
        for x in ctxs:
            x.close()


@pytest.mark.parametrize("k,c", [(31, 2), (63, 2), (5, 10)])
@pytest.mark.parametrize("gz", [False, True])
def test_count_file_over_contexts_on_different_devices(tmp_path, monkeypatch, k, c, gz):
    """mk_count_file with the contexts counting as GPUs of their own: chunk i -> context i mod n, per-GPU leaders,
    mk_merge_devices(GATHER) into ctxs[0] -- the same table as one context, and as the oracle's chunk composition."""
    import gzip
    data = _data()
    path = tmp_path / ("s.fna.gz" if gz else "s.fna")
    path.write_bytes(gzip.compress(data, 1) if gz else data)
    chunk = 2_000_000 if not gz else 400_000  # (the on-disk size decides about chunking: the .gz is smaller)
    assert os.path.getsize(path) >= chunk
    with native.Counter(k, native.ALPHABET_NT2, device=0) as one:
        native.count_file([one], path, chunk, c)
        want = one.to_dict()
    from mercat2_amd.chunker import chunk_offsets
    offs = chunk_offsets(data, chunk)
    ref = {}
    for a, b in zip(offs[:-1], offs[1:]):
        for key, n in cpu_ref.count_text(data[a:b], k, c).items():
            ref[key] = ref.get(key, 0) + n
    assert want == ref
    monkeypatch.setenv("MK_DEVICE_PER_CONTEXT", "1")
    devices = native.plan_contexts([0, 0, 0], 1)
    ctxs = [native.Counter(k, native.ALPHABET_NT2, device=d) for d in devices]
    try:
        st = native.count_file(ctxs, path, chunk, c)
        assert st["devices"] == 3 and st["contexts"] == 3 and st["chunked"] == 1 and st["split_pieces"] == 0
        assert ctxs[0].to_dict() == want
        assert all(x.rows() == 0 for x in ctxs[1:])
    finally:
        for x in ctxs:
            x.close()


@pytest.mark.parametrize("k,c", [(31, 3), (5, 10), (33, 2)])
def test_single_filter_unit_split_over_devices(tmp_path, monkeypatch, k, c):
    """A file below the chunk size is ONE filter unit (lib/mercat2_kmers.py:73-76).  On several GPUs it is counted in
    pieces cut at record starts, unfiltered, and min_count is applied to the sum: the table of find_kmers(file)."""
    reads = native.synth_reads(60_000, 7, 30_000, 150, 8).tobytes()
    # sequence lines that CONTAIN '>' (the Chunker would cut there; a piece boundary there would lose windows)
    tricky = b"".join(b">t%d\nACGTACGTAC>GTACGTACGTACGTACGTTTGACCA\nAC>GTTGCAAACCGGTTACGATCGATCGGGATATC\n" % i for i in range(3000))
    data = reads[: len(reads) // 2] + tricky + reads[len(reads) // 2:] + read_input("edge_ws.fa")
    path = tmp_path / "one_unit.fna"
    path.write_bytes(data)
    want = cpu_ref.count_text(data, k, c)
    monkeypatch.setenv("MK_DEVICE_PER_CONTEXT", "1")
    monkeypatch.setenv("MK_SPLIT_MIN", "100000")
    ctxs = [native.Counter(k, native.ALPHABET_NT2, device=0) for _ in range(3)]
    try:
        st = native.count_file(ctxs, path, 100 << 20, c)  # far below 100 MiB: not chunked
        assert st["chunked"] == 0 and st["split_pieces"] >= 3 and st["chunks"] == 1
        assert ctxs[0].to_dict() == want
    finally:
        for x in ctxs:
            x.close()
    monkeypatch.delenv("MK_DEVICE_PER_CONTEXT")
    ctxs = [native.Counter(k, native.ALPHABET_NT2, device=0) for _ in range(3)]
    try:  # contexts on ONE GPU: no split, one chunk, the same table
        st = native.count_file(ctxs, path, 100 << 20, c)
        assert st["split_pieces"] == 0 and st["contexts"] == 1
        assert ctxs[0].to_dict() == want
    finally:
        for x in ctxs:
            x.close()
