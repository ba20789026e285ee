"""Pin the oracle's restatement of merge_tsv (oracle/cpu_ref.merge_tsv_text) to outputs of the reference's own function
(tests/golden/make_report_golden.py), including the rows its streaming loop misplaces."""
import json

from conftest import GOLDEN
from oracle import cpu_ref


def _table(path):
    lines = path.read_text().split("\n")[1:-1]
    return {ln.split("\t")[0]: int(ln.split("\t")[1]) for ln in lines}


def test_merge_tsv_text_is_the_reference_s():
    idx = json.loads((GOLDEN / "report" / "transposed.json").read_text())
    for case, g in idx.items():
        tables = {name: _table(GOLDEN / rel) for name, rel in g["inputs"].items()}
        want = (GOLDEN / "report" / (case + "_merged.tsv")).read_text()
        assert cpu_ref.merge_tsv_text(tables) == want, case
    # the quirk, spelled out: b's AAAAC (12) has no row of its own, its count sits under ACGTA
    small = (GOLDEN / "report" / "small_merged.tsv").read_text()
    assert "AAAAC" not in small and "ACGTA\t11\t12\n" in small
    tables = {name: _table(GOLDEN / rel) for name, rel in idx["small"]["inputs"].items()}
    assert "AAAAC\t0\t12\n" in cpu_ref.union_tsv_text(tables)
