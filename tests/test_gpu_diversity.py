"""Alpha diversity from the table on the GPU (mk_alpha_stats + the closed forms of
mercat2_amd/diversity.py) against the metrics the reference printed for its own committed tables
(tests/golden/diversity/alpha_cases.json) and against the oracle's restatement of scikit-bio's
functions on tables with singletons and doubletons (lib/mercat2_diversity.py:13-53)."""
import json
import random
from pathlib import Path

import numpy as np
import pytest

from mercat2_amd import diversity, native
from oracle import alpha_ref, cpu_ref

pytestmark = pytest.mark.gpu
CASES = json.loads((Path(__file__).parent / "golden" / "diversity" / "alpha_cases.json").read_text())


def _table_with_counts(counts, k=12):
    """A RAW-alphabet context whose rows carry exactly these counts (distinct made-up k-mers)."""
    kmers = np.frombuffer(b"".join(b"%0*d" % (k, i) for i in range(len(counts))), dtype=np.uint8).reshape(len(counts), k)
    ctx = native.Counter(k, native.ALPHABET_RAW)
    ctx.import_exotic(kmers, np.asarray(counts, dtype=np.uint64))
    return ctx


def test_committed_metrics_from_gpu_moments():
    bad = []
    for name, case in CASES.items():
        counts = [v for v, rows in case["counts"] for _ in range(rows)]
        with _table_with_counts(counts) as ctx:
            got = diversity.alpha_from_stats(ctx.alpha_stats())
        for metric, want in case["expected"].items():
            if got[metric] != want:
                bad.append((name, metric, got[metric], want))
    assert not bad, bad[:10]


@pytest.mark.parametrize("k,alphabet,c", [(3, native.ALPHABET_NT2, 1), (21, native.ALPHABET_NT2, 1), (32, native.ALPHABET_NT2, 1),
                                          (40, native.ALPHABET_NT2, 2), (5, native.ALPHABET_AA5, 1), (9, native.ALPHABET_RAW, 1)])
def test_every_table_kind_against_the_oracle(tmp_path, k, alphabet, c):
    """Counted (not imported) tables in every mode, with singletons, doubletons and the all-T key."""
    rng = random.Random(k)
    genome = bytes(rng.choice(b"ACGT") for _ in range(3000))
    reads = b"".join(b">r%d\n" % i + genome[a:a + 100] + b"\n" for i, a in enumerate(rng.randrange(0, 2900) for _ in range(200)))
    data = reads + b">t\n" + b"T" * 70 + b"\n>n\nACGTNACGTNNACGTACGTTGCATGCATGCAACGTACGTAGCTAGCTAGCTAGCATCGATCGA\n"
    want = alpha_ref.alpha_table(list(cpu_ref.count_text(data, k, c).values()))
    with native.Counter(k, alphabet) as ctx:
        ctx.count_chunk(data, c)
        st = ctx.alpha_stats()
        assert st["observed"] == len(cpu_ref.count_text(data, k, c))
        assert diversity.alpha_from_stats(st) == want
        out = tmp_path / "alpha.tsv"
        diversity.compute_alpha_diversity("s", ctx, out)
        tsv = tmp_path / "s_counts.tsv"
        ctx.write_tsv(tsv, "s")
    assert out.read_text() == "Metric\ts\n" + "".join("%s\t%s\n" % (m, want[m]) for m in alpha_ref.METRICS)
    out2 = tmp_path / "alpha_from_tsv.tsv"       # the reference's signature: a TSV path
    diversity.compute_alpha_diversity("s", tsv, out2)
    assert out2.read_text() == out.read_text()


def test_empty_table():
    with native.Counter(5, native.ALPHABET_NT2) as ctx:
        assert set(diversity.alpha_from_stats(ctx.alpha_stats()).values()) == {"NA"}
