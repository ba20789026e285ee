"""removeN's effect on the count, on the GPU (mk_set_clean): the RAW nucleotide FASTA counted in clean mode must give the
table of the CLEANED text -- oracle: the removeN restatement (oracle/clean_ref.py, pinned to the reference's function by
tests/golden/clean_cases.json) followed by the find_kmers restatement (oracle/cpu_ref.py) -- and the GPU-derived N runs,
G + C count and length must be those of the reference's rewrite."""
import json
import re

import numpy as np
import pytest

from conftest import GOLDEN, read_input
from mercat2_amd import native
from oracle import clean_ref, cpu_ref

pytestmark = pytest.mark.gpu
CASES = json.loads((GOLDEN / "clean_cases.json").read_text())


def _want(raw: bytes, k: int, c: int, toupper: bool):
    cleaned, gc, total = clean_ref.clean_text(raw.decode("utf-8"), toupper)
    return cpu_ref.count_text(cleaned.encode(), k, c), cleaned, gc, total


def _count_clean(raw: bytes, k: int, c: int, toupper: bool):
    with native.Counter(k, native.ALPHABET_NT2) as ctx:
        ctx.set_clean(True, toupper)
        ctx.count_chunk(raw, c)
        return ctx.to_dict(), ctx.clean_stats(), ctx.clean_runs()


def _synthetic():
    rng = np.random.default_rng(11)

    def seq(n, alphabet="ACGT"):
        return "".join(alphabet[x] for x in rng.integers(0, len(alphabet), n))
    recs = ["ACGTACGTNNACGT\nGGGTTTCCCAAA\n"]  # text in front of the first header: dropped
    recs.append(">r1 plain\n" + "\n".join(seq(70) for _ in range(5)) + "\n")
    recs.append(">r2 runs across line breaks\n" + seq(50) + "NNNN\nNN" + seq(60) + "\nN\n" + seq(40) + "N\n")
    recs.append(">r3 leading and trailing runs\nNNN" + seq(100) + "NNNNN\n")
    recs.append(">r4 lower case n is no cut, lower case bases\n" + seq(40) + "nnn" + seq(40, "ACGTacgt") + "N" + seq(30, "acgt") + "\n")
    recs.append(">r5 only N\nNNNNNNNNNN\n>r6 empty\n>r7 iupac and stars\n" + seq(50, "ACGTRYKM") + "*" + seq(30) + "N" + seq(45) + "\n")
    recs.append(">r8 crlf\r\n" + seq(64) + "\r\n" + seq(10) + "NN" + seq(64) + "\r\n")
    recs.append(">r9 long single line " + "x" * 50 + "\n" + seq(900) + "N" * 120 + seq(700) + "\n")
    return "".join(recs).encode()


@pytest.mark.parametrize("k,c", [(5, 2), (21, 1), (31, 1), (33, 1), (3, 1)])
@pytest.mark.parametrize("toupper", [False, True])
def test_clean_mode_counts_what_the_cleaned_text_holds(k, c, toupper):
    raw = _synthetic()
    want, cleaned, gc, total = _want(raw, k, c, toupper)
    got, st, (starts, ends) = _count_clean(raw, k, c, toupper)
    assert got == want
    headers = [ln for ln in cleaned.split("\n") if ln.startswith(">")]
    assert st["n_runs"] == len(starts) == len(headers) - st["header_lines"]
    assert np.all(starts < ends) and np.all(ends[:-1] <= starts[1:])
    assert int(np.sum(ends - starts)) == st["n_bytes"]


@pytest.mark.parametrize("name", ["Scaffolds_with-NNN.fna.gz", "RW1.fna.gz", "edge_clean.fa"])
@pytest.mark.parametrize("toupper", [False, True])
def test_golden_inputs_in_clean_mode(name, toupper):
    """Real inputs of the reference's own runs: the table of the cleaned file, and the rewrite's own figures -- pieces,
    N bytes, G + C, total length -- derived on the GPU, against the reference function's output (clean_cases.json)."""
    raw = read_input(name)
    case = CASES.get("%s|%s" % (name, "upper" if toupper else "asis"))  # (not every input has both goldens)
    for k, c in ((5, 10), (31, 1)):
        want, cleaned, gc, total = _want(raw, k, c, toupper)
        assert case is None or len(cleaned.encode()) == case["bytes"]  # (the oracle's rewrite is the reference's)
        try:
            got, st, (starts, ends) = _count_clean(raw, k, c, toupper)
        except native.CleanUnsupported:
            assert name == "edge_clean.fa"  # (holds lines the GPU mode declines: see the next test)
            return
        assert got == want, (name, k, c)
    # pieces: every header line of the cleaned text that the rewrite numbered; runs = pieces - split records
    headers = [ln for ln in cleaned.split("\n") if ln.startswith(">")]
    raw_headers = st["header_lines"]
    assert len(headers) == raw_headers + st["n_runs"]
    assert int(np.sum(ends - starts)) == st["n_bytes"] == raw.count(b"N") - sum(h.count("N") for h in raw.decode().split("\n") if h.startswith(">"))
    if not toupper:
        # the reference's GC figure = (G + C of the sequence + of the headers of split records) / (length likewise)
        split_headers = [h for h in headers if re.match(r">\S+_\d+ ", h) and h not in raw.decode()]
        assert st["gc_count"] + sum(h.count("G") + h.count("C") for h in split_headers) == gc
        assert st["symbols"] + sum(len(h) for h in split_headers) == total
        assert case is None or 100.0 * gc / total == case["gc"]


def test_runs_are_where_split_sequenceN_cuts():
    """Run boundaries in the parsed stream (records separated by one byte): the cuts of lib/mercat2_fasta.py:35-38."""
    raw = b">a x\nACGTNNNACG\nTNAC\n>b\nNNACGT\nACGTN\n>c\nACGT\n"
    with native.Counter(3, native.ALPHABET_NT2) as ctx:
        ctx.set_clean(True, False)
        ctx.count_chunk(raw, 1)
        starts, ends = ctx.clean_runs()
        st = ctx.clean_stats()
    # parsed stream: "\nACGTNNNACGTNAC" + "\nNNACGTACGTN" + "\nACGT"
    stream = "\nACGTNNNACGTNAC\nNNACGTACGTN\nACGT"
    want = [(m.start(), m.end()) for m in re.finditer(r"N+", stream)]
    assert list(zip(starts.tolist(), ends.tolist())) == want
    assert st["n_runs"] == 4 and st["n_bytes"] == 7 and st["header_lines"] == 3
    assert st["symbols"] == len(stream) - 3 - 7 and st["gc_count"] == stream.count("G") + stream.count("C")


@pytest.mark.parametrize("raw,why", [
    (b">a\nACGT ACGTNNACGT\n", "blank"),          # textwrap would drop the blank at a line break of the split record
    (b">a\nACGT>CGTNNACGT\n", "'>'"),             # a wrapped line could start with '>' and be taken for a header
    (b">a\nACG\x7fTNNACGT\n", "0x7F"),
    (b"  >indented first header\nACGTNNACGT\n>b\nACGT\n", "front"),
])
def test_text_the_gpu_does_not_reproduce_is_refused_not_guessed(raw, why):
    with native.Counter(5, native.ALPHABET_NT2) as ctx:
        ctx.set_clean(True, False)
        with pytest.raises(native.CleanUnsupported):
            ctx.count_chunk(raw, 1)
        assert ctx.rows() == 0  # nothing of the chunk was counted
        ctx.set_clean(False)
        ctx.count_chunk(clean_ref.clean_text(raw.decode(), False)[0].encode(), 1)  # the host rewrite's text instead
        assert ctx.to_dict() == _want(raw, 5, 1, False)[0]


def test_clean_mode_is_for_nucleotides():
    with native.Counter(3, native.ALPHABET_AA5) as ctx:
        with pytest.raises(native.MercatHipError):
            ctx.set_clean(True)
