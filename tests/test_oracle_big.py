"""Pin both oracles to the reference's answers on ALL of data/5-genomes-fna_gz and data/5-genomes-faa_gz
(BASELINE configs 1 and 4 name the five genomes / proteomes): tests/golden/expected_big.json, made by
tests/golden/make_golden.py from the reference's find_kmers."""
import hashlib
import json

import numpy as np
import pytest

from conftest import GOLDEN, read_input
from oracle import c_oracle, cpu_ref

BIG = json.loads((GOLDEN / "expected_big.json").read_text())


def _by_input():
    groups = {}
    for case in BIG.values():
        groups.setdefault(case["input"], []).append(case)
    return sorted(groups.items())


@pytest.mark.parametrize("fname,cases", _by_input(), ids=[g[0] for g in _by_input()])
def test_c_oracle_on_all_five(fname, cases):
    data = read_input(fname)
    for case in cases:
        kmers, counts = c_oracle.count(data, case["k"], case["c"])
        got = {"rows": int(counts.size), "sum": int(counts.sum()),
               "keys_sha256": hashlib.sha256(kmers.tobytes()).hexdigest(),
               "counts_sha256": hashlib.sha256(counts.astype("<u8").tobytes()).hexdigest()}
        assert got == {x: case[x] for x in got}, (fname, case["k"], case["c"])


@pytest.mark.parametrize("fname,cases", _by_input(), ids=[g[0] for g in _by_input()])
def test_python_oracle_small_k_on_all_five(fname, cases):
    """The Python restatement on the small-k cases (k=3 is BASELINE config 1 / 4; large k would take minutes)."""
    for case in cases:
        if case["k"] > 5:
            continue
        table = cpu_ref.find_kmers(GOLDEN / "inputs" / fname, case["k"], case["c"])
        text = cpu_ref.tsv_text(case["basename"], table)
        assert (len(table), sum(table.values()), hashlib.sha256(text.encode()).hexdigest()) == \
               (case["rows"], case["sum"], case["sha256"]), (fname, case["k"], case["c"])
