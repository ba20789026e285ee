"""The headline workload's TABLE pinned to the reference (tests/golden/expected_s2.json, made in the build container
by tests/golden/make_s2_golden.py: the reference Chunker cuts S2, the reference find_kmers counts every chunk,
the survivors are summed as run_mercat2 sums them).  At full size, through the C ABI:
  * S2 (10 M x 150 bp, k=31, -c 10 -s 100: exactly what bench.py times), forward-strand keys: rows, sum, the sha256 of
    the TSV file mk_write_tsv writes and the digests of the exported arrays; the same through three contexts,
    mk_merge_devices and mk_write_tsv_multi (the several-GPU path);
  * the same sample with canonical keys (BASELINE config 3 as worded): the reference's per-chunk counts folded onto
    min(key, revcomp) before the filter;
  * the first two chunks of S3 (50 M-read sample, k=63) at -c 2: the two-word table at 20 M rows."""
import hashlib
import json

import numpy as np
import pytest

from conftest import GOLDEN
from mercat2_amd import native
from mercat2_amd.chunker import chunk_offsets

pytestmark = pytest.mark.gpu


def _golden(case):
    path = GOLDEN / "expected_s2.json"
    if not path.exists():
        pytest.skip("tests/golden/expected_s2.json has not been generated")
    d = json.loads(path.read_text())
    if case not in d:
        pytest.skip("no golden for " + case)
    return d[case]


def _sha_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for block in iter(lambda: f.read(1 << 22), b""):
            h.update(block)
    return h.hexdigest()


def _check_arrays(kmers, counts, want):
    assert kmers.shape[0] == want["rows"]
    assert int(counts.sum()) == want["sum"]
    assert hashlib.sha256(np.ascontiguousarray(kmers).tobytes()).hexdigest() == want["keys_sha256"]
    assert hashlib.sha256(counts.astype("<u8").tobytes()).hexdigest() == want["counts_sha256"]


@pytest.fixture(scope="module")
def s2():
    g = _golden("S2|k31|c10|s100")
    data = native.synth_reads(g["genome"], g["genome_seed"], g["reads"], g["read_len"], g["read_seed"])
    offs = chunk_offsets(data, g["chunk_mib"] << 20)
    assert [int(x) for x in offs] == g["offsets"], "cut points differ from the reference Chunker's"
    return g, data, list(zip(offs[:-1], offs[1:]))


def test_s2_headline_table_is_the_reference_table(s2, tmp_path):
    g, data, spans = s2
    view = memoryview(data)
    want = g["forward"]
    with native.Counter(g["k"], native.ALPHABET_NT2) as ctx:
        for lo, hi in spans:
            ctx.count_chunk(view[lo:hi], g["c"])  # every chunk filtered on its own (T2)
        out = tmp_path / "S2_counts.tsv"
        assert ctx.write_tsv(out, g["basename"]) == want["rows"] == 2005694
        assert _sha_file(out) == want["sha256"]
        _check_arrays(*ctx.export(), want)


def test_s2_headline_table_through_the_multi_gpu_path(s2, tmp_path):
    g, data, spans = s2
    view = memoryview(data)
    want = g["forward"]
    n = 3
    ctxs = [native.Counter(g["k"], native.ALPHABET_NT2, device=0) for _ in range(n)]
    try:
        for i, (lo, hi) in enumerate(spans):
            ctxs[i % n].count_chunk(view[lo:hi], g["c"])
        st = native.merge_devices(ctxs, native.MERGE_RANGES | native.MERGE_BALANCED)
        assert st["rows_out"] == want["rows"]
        assert st["max_owned"] < 0.45 * want["rows"]
        out = tmp_path / "S2_counts.tsv"
        assert native.write_tsv_multi(ctxs, out, g["basename"]) == want["rows"]
        assert _sha_file(out) == want["sha256"]
    finally:
        for x in ctxs:
            x.close()


def test_s2_canonical_table_is_the_folded_reference_table(s2, tmp_path):
    g, data, spans = s2
    view = memoryview(data)
    want = g["canonical"]
    with native.Counter(g["k"], native.ALPHABET_NT2, canonical=True) as ctx:
        for lo, hi in spans:
            ctx.count_chunk(view[lo:hi], g["c"])
        out = tmp_path / "S2_counts.tsv"
        assert ctx.write_tsv(out, g["basename"]) == want["rows"]
        assert _sha_file(out) == want["sha256"]
        _check_arrays(*ctx.export(), want)


def test_s1_config2_table_is_the_reference_table(tmp_path):
    """BASELINE config 2 at full size (S1: 1 M x 150 bp from a 1 Mbp genome, k = 21, -c 10 -s 100: two chunks), one and
    two contexts: the reference's table by sha256."""
    g = _golden("S1|k21|c10|s100")
    data = native.synth_reads(g["genome"], g["genome_seed"], g["reads"], g["read_len"], g["read_seed"])
    offs = [int(x) for x in chunk_offsets(data, g["chunk_mib"] << 20)]
    assert offs == g["offsets"] and len(offs) == 3
    view = memoryview(data)
    want = g["forward"]
    with native.Counter(g["k"], native.ALPHABET_NT2) as ctx, native.Counter(g["k"], native.ALPHABET_NT2) as other:
        for lo, hi in zip(offs[:-1], offs[1:]):
            ctx.count_chunk(view[lo:hi], g["c"])
        out = tmp_path / "S1_counts.tsv"
        assert ctx.write_tsv(out, g["basename"]) == want["rows"]
        assert _sha_file(out) == want["sha256"]
        _check_arrays(*ctx.export(), want)
        # chunk 0 in one context, chunk 1 in another, summed on the device (what bench.py's config 2 leg does)
        ctx.reset()
        ctx.count_chunk(view[offs[0]:offs[1]], g["c"])
        other.count_chunk(view[offs[1]:offs[2]], g["c"])
        ctx.merge_from(other)
        _check_arrays(*ctx.export(), want)


def test_s3_first_chunks_two_word_table_is_the_reference_table(tmp_path):
    g = _golden("S3head|k63|c2|s100|chunks2")
    data = native.synth_reads(g["genome"], g["genome_seed"], g["reads"], g["read_len"], g["read_seed"])
    offs = [int(x) for x in chunk_offsets(data, g["chunk_mib"] << 20)]
    assert offs[: len(g["offsets"])] == g["offsets"]
    view = memoryview(data)
    want = g["forward"]
    with native.Counter(g["k"], native.ALPHABET_NT2) as ctx:
        for lo, hi in zip(g["offsets"][:-1], g["offsets"][1:]):
            ctx.count_chunk(view[lo:hi], g["c"])
        assert ctx.stats()["mode_name"] == "hash128"
        out = tmp_path / "S3_counts.tsv"
        assert ctx.write_tsv(out, g["basename"]) == want["rows"]
        assert _sha_file(out) == want["sha256"]
        _check_arrays(*ctx.export(), want)
