"""removeN (lib/mercat2_fasta.py:53-119) through the native rewrite (mk_remove_n) and mercat2_amd.fasta, against
outputs of the reference's own function (tests/golden/make_clean_golden.py; for the five genomes these are also
the reference's committed results/2023-11-29/fna-5genomes_gz-10/clean/*_clean.fna.gz, byte for byte)."""
import gzip
import hashlib
import json
import os

import pytest

from conftest import GOLDEN, read_input
from mercat2_amd import fasta, native
from oracle import clean_ref

CASES = json.loads((GOLDEN / "clean_cases.json").read_text())


@pytest.mark.parametrize("case", sorted(CASES), ids=sorted(CASES))
def test_clean_text_matches_reference(case):
    c = CASES[case]
    text, stats = fasta.clean_text(read_input(c["input"]), c["toupper"])
    assert len(text) == c["bytes"] and text.count(b"\n") == c["lines"]
    assert hashlib.sha256(text).hexdigest() == c["sha256"]
    assert stats["GC Content"] == c["gc"]  # same integers, same division
    small = GOLDEN / "clean" / ("%s_%s.fna" % (c["out_name"][:-len("_clean.fna.gz")], "upper" if c["toupper"] else "asis"))
    if small.exists():
        assert text == small.read_bytes()


@pytest.mark.parametrize("case", ["RW1.fna.gz|asis", "Scaffolds_with-NNN.fna.gz|upper", "edge_clean.fa|asis", "edge_clean_odd.fa|upper"])
def test_removeN_file_like_the_reference(case, tmp_path):
    """Same path, same decompressed bytes, same size on disk (the size decides whether the sample is chunked,
    bin/mercat2.py:101: the file is written with the same gzip settings)."""
    c = CASES[case]
    out, stats = fasta.removeN(GOLDEN / "inputs" / c["input"], tmp_path / "clean", c["toupper"])
    assert out == (tmp_path / "clean" / c["out_name"]).absolute()
    assert hashlib.sha256(gzip.open(out, "rb").read()).hexdigest() == c["sha256"]
    assert os.stat(out).st_size == c["gz_size"]
    assert stats == {"GC Content": c["gc"]}


def test_native_path_handles_every_golden_input():
    """No input is handed to a second implementation: the native rewrite does all of them, the one with word breaks
    inside split sequences (edge_clean_odd.fa: blanks, tabs, hyphens -- textwrap's word rules) included."""
    for case, c in CASES.items():
        text, st = native.remove_n(read_input(c["input"]), c["toupper"])
        assert st["unsupported_record"] == -1, case
        assert hashlib.sha256(text).hexdigest() == c["sha256"], case
    assert not hasattr(fasta, "_clean_text_py") and not hasattr(fasta, "split_sequenceN")


@pytest.mark.parametrize("case", sorted(CASES), ids=sorted(CASES))
def test_oracle_restatement_is_pinned_to_the_reference(case):
    c = CASES[case]
    text, gc, total = clean_ref.clean_text(read_input(c["input"]).decode("utf-8"), c["toupper"])
    assert hashlib.sha256(text.encode()).hexdigest() == c["sha256"]
    assert 100.0 * gc / total == c["gc"]


def test_textwrap_restatement_equals_the_standard_library():
    """mk_textwrap (what mk_remove_n applies to the pieces of a split sequence) against textwrap.wrap itself on random
    ASCII text full of what textwrap cares about: blanks, tabs, other white space, hyphens and dashes between letters,
    digits and punctuation, words longer than a line."""
    import random
    import textwrap
    rng = random.Random(4)
    alphabets = ["ACGT -", "ACGTacgt \t-", "AC-G--T---x1 .,!?\"'&_", "ab-cd- e--f g\x0b\x0c\x1c\x1f\t", "A-", "-", " ", "ACGT"]
    for case in range(3000):
        ab = rng.choice(alphabets)
        n = rng.choice([0, 1, 2, 5, 30, 79, 80, 81, 82, 160, 161, 400])
        text = "".join(rng.choice(ab) for _ in range(n))
        if rng.random() < 0.3:  # long runs without a break
            at = rng.randrange(0, len(text) + 1)
            text = text[:at] + "".join(rng.choice("ACGT-") for _ in range(rng.randrange(60, 200))) + text[at:]
        for width in (80, 7, 1):
            want = [ln.encode() for ln in textwrap.wrap(text, width)]
            got = native.textwrap_lines(text.encode(), width)
            assert got == want, (case, width, text)


def test_non_ascii_bytes():
    """Header lines may hold any bytes (where the reference works on characters -- strip, split, len -- UTF-8 is
    understood); in a sequence line a byte >= 0x80 is refused, as the counting engine refuses it."""
    head = ">s\u00e9q \u00a0 d\u00e9sc\u2003x\u00a0"
    text = (head + "\nACGTNNACGT\nGG\n>plain h\u00e9\nACGT\n").encode("utf-8")
    want, gc, total = clean_ref.clean_text(text.decode("utf-8"), False)
    got, st = native.remove_n(text, False)
    assert got == want.encode("utf-8")
    assert (st["gc_count"], st["total_length"]) == (gc, total)
    with pytest.raises(native.NonAsciiInput):
        native.remove_n(">a\nACG\u00e9T\n".encode("utf-8"), False)
    # a header in another encoding is copied as it stands (the reference would fail to decode the file)
    latin = b">caf\xe9 x\nACGT\n"
    assert native.remove_n(latin, False)[0] == latin


def test_cr_only_files_take_linear_time():
    import time
    rec = b">r\r" + b"ACGT" * 20 + b"\r"
    data = rec * 200_000  # 17 MB, no '\\n' anywhere
    t0 = time.perf_counter()
    out, st = native.remove_n(data, False)
    assert time.perf_counter() - t0 < 5.0
    assert st["records"] == 200_000 and out.count(b"\n") == 400_000 and b"\r" not in out


def test_reference_errors_are_kept():
    with pytest.raises(ZeroDivisionError):
        fasta.clean_text(b"", False)
    with pytest.raises(ZeroDivisionError):
        fasta.clean_text(b"no header at all\nACGT\n", False)
    with pytest.raises(IndexError):
        fasta.clean_text(b">\nACGTNNACGT\n", False)  # header.split()[0] of an empty header


def test_committed_clean_file_is_reproduced():
    """RW1.fna.gz -> RW1_clean.fna.gz, the pair the reference committed (the other four: digests in clean.json)."""
    text, _ = fasta.clean_text(read_input("RW1.fna.gz"), False)
    assert text == read_input("RW1_clean.fna.gz")
    want = json.loads((GOLDEN / "clean.json").read_text())
    for name, w in want.items():
        got, _ = fasta.clean_text(read_input(name), False)
        assert (len(got), hashlib.sha256(got).hexdigest()) == (w["bytes"], w["sha256"]), name


def test_native_rewrite_equals_the_oracle_restatement_on_random_text():
    """Differential run: mk_remove_n against the oracle's restatement (oracle/clean_ref.py, pinned to the reference's
    function by the goldens above) on random FASTA-like text: N runs anywhere, blanks and hyphens inside sequences,
    \\n / \\r\\n / lone \\r line ends, blank and blank-padded lines, headers with several words, text before the first
    header, empty records, a missing final newline."""
    import random
    rng = random.Random(20261004)
    alphabet = "ACGTacgtNNNnRY*"
    for case in range(400):
        parts = []
        if rng.random() < 0.3:
            parts.append("".join(rng.choice(alphabet) for _ in range(rng.randrange(0, 30))) + "\n")
        for r in range(rng.randrange(0, 6)):
            words = [("w%d" % rng.randrange(100)) for _ in range(rng.randrange(1, 4))]
            sep = rng.choice([" ", "  ", "\t", " \t "])
            head = ">" + sep.join(words) + rng.choice(["", " ", "\t"])
            if rng.random() < 0.1:
                head = "  " + head
            lines = [head]
            for _ in range(rng.randrange(0, 5)):
                n = rng.choice([0, 1, 7, 60, 80, 81, 200])
                body = "".join(rng.choice(alphabet + (" -\t>" if case % 4 == 0 else "")) for _ in range(n))
                if rng.random() < 0.2:
                    body = "N" * rng.randrange(1, 90) + body
                if rng.random() < 0.2:
                    body = body + "N" * rng.randrange(1, 5)
                lines.append(rng.choice(["", " ", "\t"]) + body + rng.choice(["", " ", "  "]))
            eol = rng.choice(["\n", "\n", "\r\n", "\r"])
            parts.append(eol.join(lines) + (eol if rng.random() < 0.9 else ""))
        text = "".join(parts)
        for up in (False, True):
            try:
                want, gc, total = clean_ref.clean_text(text, up)
            except IndexError:
                with pytest.raises(IndexError):
                    native.remove_n(text.encode(), up)
                continue
            got, st = native.remove_n(text.encode(), up)
            assert st["unsupported_record"] == -1
            assert got == want.encode(), (case, up, text)
            assert (st["gc_count"], st["total_length"]) == (gc, total), (case, up, text)


def test_clean_gz_writer_streams_and_decides_early(tmp_path):
    """<base>_clean.fna.gz is written in slices -- same bytes as one gzip.open(...).write(...) but for the time stamp --
    and the chunking decision (its size against -s MiB, bin/mercat2.py:101) is out as soon as the stream has passed
    the limit, not when the file is complete; below the limit it is out when the file is."""
    import gzip
    import random
    from mercat2_amd import fasta
    random.seed(5)
    body = bytes(random.choice(b"ACGT") for _ in range(2_500_000))
    text = b">x some text\n" + b"\n".join(body[i:i + 80] for i in range(0, len(body), 80)) + b"\n"
    ref = tmp_path / "ref" / "x_clean.fna.gz"
    ref.parent.mkdir()
    with gzip.open(ref, "wb") as w:  # what the reference's writer does (text mode: one flush at close)
        w.write(text)
        w.flush()
    seen = []

    class Spy(fasta.GzDecision):
        def _grew(self, size, complete):
            was = self.chunked
            super()._grew(size, complete)
            if was is None and self.chunked is not None:
                seen.append((size, complete))
    out = tmp_path / "out" / "x_clean.fna.gz"
    out.parent.mkdir()
    d = Spy(200_000)
    size = fasta._write_clean_gz(out, text, decision=d)
    a, b = ref.read_bytes(), out.read_bytes()
    assert size == len(b) == len(a) and a[:4] == b[:4] and a[8:] == b[8:]  # (bytes 4..7: the time stamp)
    assert gzip.decompress(b) == text
    assert d.wait(0) is True and seen and seen[0][1] is False and 200_000 <= seen[0][0] < size // 2
    d2 = fasta.GzDecision(size + 1)
    assert fasta._write_clean_gz(out, text, decision=d2) == size and d2.wait(0) is False
    d3 = fasta.GzDecision(0)  # -s 0: never chunked
    assert fasta._write_clean_gz(out, text, decision=d3) == size and d3.wait(0) is False
