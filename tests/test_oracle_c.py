"""Pin the C oracle (oracle/kmer_oracle.c) to the reference's golden vectors and to the Python oracle."""
import hashlib
import json

import pytest

from conftest import GOLDEN, read_input
from oracle import c_oracle, cpu_ref

EXPECTED = json.loads((GOLDEN / "expected.json").read_text())


def _by_input():
    groups = {}
    for case in EXPECTED.values():
        groups.setdefault(case["input"], []).append(case)
    return sorted(groups.items())


@pytest.mark.parametrize("fname,cases", _by_input(), ids=[g[0] for g in _by_input()])
def test_c_oracle_matches_reference_goldens(fname, cases):
    data = read_input(fname)
    for case in cases:
        kmers, counts = c_oracle.count(data, case["k"], case["c"])
        k = case["k"]
        flat = kmers.tobytes().decode("ascii")
        text = "k-mer\t%s_Count\n" % case["basename"] + "".join(
            "%s\t%d\n" % (flat[i * k:(i + 1) * k], int(c)) for i, c in enumerate(counts))
        got = {"rows": int(counts.size), "sum": int(counts.sum()), "sha256": hashlib.sha256(text.encode()).hexdigest()}
        assert got == {x: case[x] for x in ("rows", "sum", "sha256")}, (fname, k, case["c"])


def test_c_oracle_equals_python_oracle_on_odd_text():
    data = (GOLDEN / "inputs" / "edge_ws.fa").read_bytes() + b"\r\n \t>x y\r AC*GT\x0b\n\nNN>NN\n"
    for k in (1, 2, 7, 31):
        assert c_oracle.count_dict(data, k, 1) == cpu_ref.count_text(data, k, 1)
