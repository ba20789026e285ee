"""CPU-side checks of the drop-in boundary: the C ABI library loads and exports what the header
declares, the host helpers behind it (virtual Chunker, synthetic reads) are exact, and the
product path fails loudly without a GPU.  No compute kernels are launched here."""
import ctypes
import hashlib
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, read_input
from mercat2_amd import native
from mercat2_amd.chunker import Chunker, chunk_offsets, human2bytes, _normalise_newlines
from oracle import cpu_ref

HEADER = ROOT / "include" / "mercat_hip.h"


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(mk_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(native.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = native.lib()
    for name in declared_functions():
        assert getattr(lib, name) is not None, name
    assert b"gfx950" in lib.mk_version()


def test_stats_struct_matches_c_layout(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "mercat_hip.h"\nint main(){printf("%zu",sizeof(mk_stats_t));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", str(ROOT / "include"), str(src), "-o", str(exe)])
    assert int(subprocess.check_output([str(exe)])) == ctypes.sizeof(native.Stats)


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="only meaningful on a box without a GPU")
def test_no_gpu_means_loud_failure_not_cpu_fallback():
    with pytest.raises(native.MercatHipError) as e:
        native.Counter(31)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)
    from mercat2_amd.kmers import find_kmers
    with pytest.raises(native.MercatHipError):
        find_kmers(GOLDEN / "inputs" / "A.fasta", 31, 1)


def test_missing_library_is_an_error():
    code = ("import os,sys; os.environ['MERCAT_HIP_LIB']='/nonexistent/libmercat_hip.so'; sys.path.insert(0, %r);"
            "from mercat2_amd import native\n"
            "try:\n native.lib()\nexcept native.MercatHipError as e:\n print('LOUD', e.code)\n") % str(ROOT)
    out = subprocess.check_output([sys.executable, "-c", code]).decode()
    assert "LOUD -2" in out


def test_product_package_never_imports_the_oracle():
    for p in (ROOT / "mercat2_amd").rglob("*.py"):
        text = p.read_text()
        assert "oracle" not in re.sub(r'""".*?"""', "", text, flags=re.S).replace("# ", ""), p
    for p in (ROOT / "mercat2_amd" / "csrc").iterdir():
        if p.suffix in (".hip", ".cpp", ".h"):
            assert "cpu_ref" not in p.read_text(), p


# ------------------------------------------------------------------------- virtual Chunker
def test_chunk_cuts_match_reference_chunker(chunk_golden):
    """Cut points of the reference Chunker (lib/mercat2_Chunker.py:39-59) on every golden case:
    same number of chunks and byte-identical (newline-normalised) chunk contents."""
    for name, g in chunk_golden["chunks"].items():
        data = read_input(g["input"])
        offs = chunk_offsets(data, g["bytes"])
        got = [hashlib.sha256(_normalise_newlines(data[a:b])).hexdigest() for a, b in zip(offs[:-1], offs[1:])]
        assert got == g["sha256"], name
        norm = [0]
        for a, b in zip(offs[:-1], offs[1:]):
            norm.append(norm[-1] + len(_normalise_newlines(data[a:b])))
        assert norm[:-1] == g["offsets"] and norm[-1] == g["total"], name


def test_chunker_class_writes_reference_files(chunk_golden, inputs_dir, tmp_path):
    for name, g in chunk_golden["chunks"].items():
        dest = tmp_path / name.replace("|", "_")
        c = Chunker(str(inputs_dir / g["input"]), str(dest), g["size"], ">")
        files = sorted(c.files)
        assert [Path(f).name for f in files] == g["names"], name
        assert [hashlib.sha256(Path(f).read_bytes()).hexdigest() for f in files] == g["sha256"], name


def test_chunk_cuts_against_oracle_on_random_text():
    rng = np.random.default_rng(5)
    alphabet = np.frombuffer(b"ACGT>\n\r \tN*", dtype=np.uint8)
    for trial in range(40):
        n = int(rng.integers(0, 3000))
        data = alphabet[rng.integers(0, len(alphabet), n)].tobytes()
        size = int(rng.integers(0, 400))
        offs = chunk_offsets(data, size)
        import io
        groups = cpu_ref.split_lines(io.TextIOWrapper(io.BytesIO(data), encoding="utf-8", newline=None), size)
        want = ["".join(g).encode() for g in groups]
        got = [_normalise_newlines(data[a:b]) for a, b in zip(offs[:-1], offs[1:])]
        assert got == want, (trial, size)


def test_human2bytes(chunk_golden):
    for s, want in chunk_golden["human2bytes"].items():
        assert human2bytes(s) == want
    with pytest.raises(ValueError):
        human2bytes("12 foo")


# ------------------------------------------------------------------------ synthetic reads
M64 = (1 << 64) - 1


def _splitmix(state):
    state = (state + 0x9E3779B97F4A7C15) & M64
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return state, z ^ (z >> 31)


def py_synth(glen, gseed, reads, rlen, rseed, sub_ppm=0, first=0):
    """Pure-Python mirror of mk_synth_reads (mercat2_amd/csrc/mk_host.cpp)."""
    g, s = [], gseed
    while len(g) < glen:
        s, r = _splitmix(s)
        g.extend((r >> (2 * j)) & 3 for j in range(32))
    g = g[:glen]
    out = []
    for i in range(first, first + reads):
        s = (rseed + i * 0x632BE59BD9B4E019) & M64
        s, r0 = _splitmix(s)
        start = r0 % (glen - rlen + 1)
        s, r1 = _splitmix(s)
        codes = g[start:start + rlen]
        if r1 >> 63:
            codes = [3 - c for c in reversed(codes)]
        if sub_ppm:
            for j in range(rlen):
                s, d = _splitmix(s)
                if d % 1000000 < sub_ppm:
                    codes[j] = (codes[j] + 1 + ((d >> 32) % 3)) & 3
        out.append(">r%d\n%s\n" % (i, "".join("ACGT"[c] for c in codes)))
    return "".join(out).encode()


def test_synth_reads_match_python_mirror():
    for args in [(1000, 1, 50, 20, 2, 0, 0), (777, 9, 30, 150, 4, 0, 95), (5000, 3, 40, 100, 7, 20000, 3)]:
        assert native.synth_reads(*args).tobytes() == py_synth(*args)


def test_synth_reads_slices_are_consistent():
    whole = native.synth_reads(2000, 5, 15, 30, 6).tobytes()
    part = native.synth_reads(2000, 5, 10, 30, 6, 0, 5).tobytes()
    assert whole.endswith(part)
