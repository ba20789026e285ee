"""The planning side of the several-GPU path, on the CPU box: which chunk goes to which device, who owns which key
range, where one filter unit is cut into pieces, and the layout of the structs the binding shares with the library."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from conftest import read_input
from mercat2_amd import native

ROOT = Path(__file__).resolve().parent.parent


def test_owner_bounds_are_equal_key_ranges():
    for bits in (6, 42, 62, 64):
        for n in (1, 2, 3, 7, 8):
            b = native.owner_bounds(bits, n)
            assert len(b) == n - 1 and b == sorted(b)
            assert b == [((i << bits) + n - 1) // n for i in range(1, n)]
            # owner of a key = bounds <= key: every key of [0, 2^bits) has one owner, ranges differ by at most one key
            sizes = np.diff([0] + b + [1 << bits])
            assert sizes.min() >= 0 and sizes.max() - sizes.min() <= 1


def test_owner_bounds_match_the_torchrun_path():
    dist = pytest.importorskip("mercat2_amd.dist")
    for bits in (42, 62, 64):
        for n in (2, 4, 8):
            assert native.owner_bounds(bits, n) == dist.range_bounds(bits, n)


def test_chunk_to_device_map():
    """mk_count_file deals chunk i to ctxs[i mod nctx]; mk_plan_contexts orders the contexts so that this is
    device devices[i mod ndev] and consecutive chunks of one device use its streams in turn."""
    devices, streams = [0, 1, 2, 3, 4, 5, 6, 7], 2
    ctx_dev = native.plan_contexts(devices, streams)
    assert len(ctx_dev) == 16
    per_device = {}
    for chunk in range(80):
        j = chunk % len(ctx_dev)
        assert ctx_dev[j] == devices[chunk % len(devices)]
        per_device.setdefault(ctx_dev[j], []).append(j)
    for d, js in per_device.items():
        assert len(js) == 10 and len(set(js)) == streams  # both streams of the device take turns
        assert all(a != b for a, b in zip(js, js[1:]))
    assert native.plan_contexts([2, 5], 1) == [2, 5]
    assert native.plan_contexts([3], 3) == [3, 3, 3]


def _py_record_cuts(text: bytes, piece: int):
    """Restatement: walk text-mode lines; a line whose first non-blank byte is '>' opens the next piece once the
    piece holds >= `piece` newline-normalised bytes."""
    cuts, written, pos, n = [], 0, 0, len(text)
    blanks = b" \t\n\x0b\x0c\r\x1c\x1d\x1e\x1f"
    while pos < n:
        m = re.compile(rb"\r\n|\n|\r").search(text, pos)
        end, after = (m.start(), m.end()) if m else (n, n)
        line = text[pos:end]
        if written >= piece and line.lstrip(blanks)[:1] == b">":
            cuts.append(pos)
            written = 0
        written += len(line) + (1 if m else 0)
        pos = after
    return cuts


@pytest.mark.parametrize("name", ["edge_ws.fa", "edge_lengths.fa", "edge_reads.fna", "A.fasta"])
def test_record_cuts(name):
    text = read_input(name)
    for piece in (1, 100, 1000, 5000):
        want = _py_record_cuts(text, piece)
        for block in (1, 7, 64, 1 << 20):
            got = native.record_cuts(text, piece, block).tolist()
            assert got == want, (name, piece, block)
        # every cut is a record start (never a sequence line that merely contains '>'), and never looser than the Chunker
        chunker = set(native.stream_cuts(text, piece, 1 << 20).tolist())
        for c in want:
            assert text[c:].lstrip(b" \t\x0b\x0c\x1c\x1d\x1e\x1f")[:1] == b">"
            assert c == 0 or text[c - 1:c] in (b"\n", b"\r")
        if name != "edge_ws.fa":
            assert set(want) <= chunker or piece == 1


def test_record_cuts_skip_lines_that_only_contain_the_delimiter():
    text = b">a\nACGT\nAC>GT\n>b\nAAAA\n  >c\nCCCC\n\t>d x\nGG\n"
    assert native.stream_cuts(text, 1, 3).tolist() == [8, 14, 22, 32]
    assert native.record_cuts(text, 1, 3).tolist() == [14, 22, 32]


def _c_struct_fields(name):
    """Field list of a typedef struct in include/mercat_hip.h: [(ctype, field), ...] with arrays as (type, name, len)."""
    text = (ROOT / "include" / "mercat_hip.h").read_text()
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for stmt in body.split(";"):
        stmt = " ".join(stmt.split())
        if not stmt:
            continue
        ctype, rest = stmt.split(" ", 1)
        for f in rest.split(","):
            f = f.strip()
            m = re.match(r"(\w+)\[(\d+)\]", f)
            out.append((ctype, m.group(1), int(m.group(2))) if m else (ctype, f, 0))
    return out


@pytest.mark.parametrize("cname,cls", [("mk_merge_stats_t", "MergeStats"), ("mk_file_stats_t", "FileStats"), ("mk_stats_t", "Stats"),
                                       ("mk_clean_stats_t", "CleanStats"), ("mk_alpha_t", "AlphaStats"), ("mk_export_stats_t", "ExportStats")])
def test_struct_layouts_match_the_header(cname, cls):
    ctype = {"uint64_t": C.c_uint64, "int64_t": C.c_int64, "int32_t": C.c_int32, "double": C.c_double}
    want = _c_struct_fields(cname)
    got = getattr(native, cls)._fields_
    assert [f[1] for f in want] == [g[0] for g in got]
    for (t, _, arr), (_, gt) in zip(want, got):
        assert gt is (ctype[t] * arr if arr else ctype[t]) or (arr and gt._type_ is ctype[t] and gt._length_ == arr)


def test_abi_number_is_checked():
    assert native.lib().mk_version().decode().split()[1].split(".")[0] == str(native.MK_ABI)
