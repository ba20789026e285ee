"""The streaming Chunker scanner behind mk_count_file (mercat2_amd/csrc/mk_cutscan.h) against the
whole-buffer statement of the same rule (mk_chunk_cuts) and the Python oracle's chunker, over
texts built to hit every boundary case: "\\r\\n" and lone "\\r" line ends (also split across blocks),
'>' inside lines, unterminated last lines, very long lines, thresholds that land exactly on a line
start.  No GPU needed (lib/mercat2_Chunker.py:39-59 is the rule)."""
import io
import random

import numpy as np
import pytest

from mercat2_amd import native
from oracle import cpu_ref


def _text(rng, nlines, style):
    out = []
    for i in range(nlines):
        r = rng.random()
        if style == "reads":
            line = b">r%d" % i if i % 2 == 0 else bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 40)))
        elif r < 0.15:
            line = b">" + bytes(rng.choice(b"abc xyz") for _ in range(rng.randint(0, 12)))
        elif r < 0.25:
            line = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 6))) + b">" + b"tail" * rng.randint(0, 3)
        elif r < 0.3:
            line = b""
        elif r < 0.33:
            line = bytes(rng.choice(b"ACGTN") for _ in range(rng.randint(200, 900)))
        else:
            line = bytes(rng.choice(b"ACGT*") for _ in range(rng.randint(0, 70)))
        if style == "lf":
            term = b"\n"
        elif style == "crlf":
            term = b"\r\n"
        elif style == "cr":
            term = b"\r"
        elif style == "reads":
            term = b"\n"
        else:
            term = rng.choice([b"\n", b"\n", b"\r\n", b"\r"])
        out.append(line + term)
    text = b"".join(out)
    if rng.random() < 0.5 and text:
        text = text.rstrip(b"\r\n") + rng.choice([b"", b">last", b"ACGT", b"\r"])
    return text


@pytest.mark.parametrize("style", ["lf", "crlf", "cr", "mixed", "reads"])
def test_stream_cuts_match_whole_buffer_rule(style):
    rng = random.Random(hash(style) & 0xFFFF)
    for case in range(60):
        text = _text(rng, rng.randint(0, 120), style)
        for chunksize in (1, 7, 64, 300, 1500, 10 ** 9):
            want = native.chunk_cuts(text, chunksize)
            for block in (1, 2, 3, 17, 64, 257, 4096, 1 << 20):
                got = native.stream_cuts(text, chunksize, block)
                assert got.tolist() == want.tolist(), (style, case, chunksize, block, text[:200])


def test_stream_cuts_match_python_oracle():
    rng = random.Random(5)
    for case in range(40):
        text = _text(rng, rng.randint(1, 80), "mixed")
        for chunksize in (5, 100, 700):
            groups = cpu_ref.split_lines(io.TextIOWrapper(io.BytesIO(text), encoding="utf-8", newline=None), chunksize)
            want = ["".join(g).encode() for g in groups]
            offs = [0] + native.stream_cuts(text, chunksize, 13).tolist() + [len(text)]
            got = [text[a:b].replace(b"\r\n", b"\n").replace(b"\r", b"\n") for a, b in zip(offs[:-1], offs[1:])]
            assert got == want, (case, chunksize)


def test_threshold_exactly_on_a_line_start():
    text = b">a\nACGT\n>b\nACGT\n>c\nACGT\n"      # 8 bytes per record
    for block in (1, 5, 8, 64):
        assert native.stream_cuts(text, 8, block).tolist() == [8, 16]
        assert native.stream_cuts(text, 9, block).tolist() == [16]
        assert native.stream_cuts(text, 16, block).tolist() == [16]
    crlf = text.replace(b"\n", b"\r\n")           # 10 raw bytes per record, still 8 written
    for block in (1, 3, 10, 64):
        assert native.stream_cuts(crlf, 8, block).tolist() == [10, 20]
        assert native.stream_cuts(crlf, 9, block).tolist() == [20]


def test_long_line_that_may_cut_is_held_across_blocks():
    text = b">h\n" + b"A" * 50 + b"\n" + b"C" * 5000 + b">x" + b"G" * 5000 + b"\n>t\nAC\n"
    want = native.chunk_cuts(text, 20).tolist()
    assert want == [54, 54 + 10003]
    for block in (7, 100, 4096):
        assert native.stream_cuts(text, 20, block).tolist() == want


def test_empty_and_degenerate():
    assert native.stream_cuts(b"", 10, 4).tolist() == []
    assert native.stream_cuts(b"\n\n\n", 1, 1).tolist() == []
    assert native.stream_cuts(b">", 0, 1).tolist() == native.chunk_cuts(b">", 0).tolist()
