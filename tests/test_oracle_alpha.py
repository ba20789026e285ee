"""oracle/alpha_ref.py against every (sample table, printed alpha metrics) pair the reference committed
(tests/golden/diversity/alpha_cases.json, built by tests/golden/make_diversity_golden.py)."""
import json
from pathlib import Path

from oracle import alpha_ref

CASES = json.loads((Path(__file__).parent / "golden" / "diversity" / "alpha_cases.json").read_text())


def _expand(pairs):
    out = []
    for value, rows in pairs:
        out.extend([value] * rows)
    return out


def test_alpha_oracle_reproduces_the_committed_metrics():
    assert len(CASES) >= 90
    bad = []
    for name, case in CASES.items():
        got = alpha_ref.alpha_table(_expand(case["counts"]))
        for metric, want in case["expected"].items():
            if got[metric] != want:
                bad.append((name, metric, got[metric], want))
    assert not bad, bad[:10]


def test_alpha_oracle_singleton_branches_are_consistent():
    counts = [1] * 40 + [2] * 10 + [3] * 5 + [50, 70, 11, 10]
    t = alpha_ref.alpha_table(counts)
    lo, hi = [float(x) for x in t["chao1_ci"].strip("[]").split(", ")]
    assert lo <= float(t["chao1"]) <= hi and float(t["chao1"]) > len(counts)
    assert float(t["ace"]) > len(counts) and 0 < float(t["goods_coverage"]) < 1
    assert alpha_ref.alpha_table([1, 1, 1])["ace"] == "NA" and alpha_ref.alpha_table([1, 1, 1])["fisher_alpha"] == "NA"
