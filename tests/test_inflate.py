"""The file reader's gzip decoder (mercat2_amd/csrc/mk_inflate.h, through mk_gunzip) against zlib:
every block type (stored, fixed, dynamic), every compression level and strategy zlib offers, texts from
empty to long runs and incompressible bytes, output handed over in blocks from 1 byte up (matches and
stored blocks cut by block ends), several members, padding, and damaged files.  No GPU needed."""
import gzip
import os
import random
import zlib

import pytest

from mercat2_amd import native


def _gz(data: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=31, memlevel=8) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
    return c.compress(data) + c.flush()


def _texts():
    rng = random.Random(7)
    dna = bytes(rng.choice(b"ACGT") for _ in range(300_000))
    reads = b"".join(b">r%d\n" % i + dna[a:a + 150] + b"\n" for i, a in enumerate(rng.randrange(0, 299_000) for _ in range(3000)))
    return {
        "empty": b"",
        "one": b"A",
        "short": b"hello, hello, hello\n",
        "runs": b"A" * 100_000 + b"\n" + b"ACGT" * 30_000 + b"N" * 70_000,
        "reads": reads,
        "random": random.Random(9).randbytes(200_000),
        "text": (b"the quick brown fox jumps over the lazy dog; " * 4000) + dna[:5000],
        "far": dna[:40_000] + random.Random(11).randbytes(20_000) + dna[:40_000],   # matches at distances up to 32 KiB
    }


TEXTS = _texts()


@pytest.mark.parametrize("name", list(TEXTS))
def test_every_level_and_strategy(name):
    data = TEXTS[name]
    for level in (0, 1, 3, 6, 9):
        for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
            gz = _gz(data, level, strategy)
            for block in (1 << 22, 4096, 257):
                got, members = native.gunzip(gz, len(data), block)
                assert got == data and members == 1, (name, level, strategy, block)


def test_tiny_output_blocks_cut_matches_and_stored_blocks():
    data = TEXTS["runs"][:20_000] + TEXTS["reads"][:20_000]
    for level in (0, 1, 9):
        gz = _gz(data, level)
        for block in (1, 2, 3, 7, 31):
            assert native.gunzip(gz, len(data), block)[0] == data, (level, block)


def test_small_memlevel_makes_many_small_dynamic_blocks():
    data = TEXTS["reads"]
    assert native.gunzip(_gz(data, 6, memlevel=1), len(data))[0] == data


def test_members_padding_and_header_fields(tmp_path):
    a, b = TEXTS["reads"][:50_000], TEXTS["text"][:70_000]
    p = tmp_path / "named.txt.gz"
    with gzip.GzipFile(p, "wb", mtime=0) as fh:     # FNAME in the header
        fh.write(a)
    named = p.read_bytes()
    extra = bytearray(_gz(b))
    extra[3] |= 4 | 16                               # FEXTRA + FCOMMENT
    extra[10:10] = b"\x05\x00hello" + b"a comment\x00"
    gz = named + bytes(extra) + b"\0" * 100 + gzip.compress(b"") + b"\0"
    got, members = native.gunzip(gz, len(a) + len(b), 5000)
    assert got == a + b and members == 3
    assert gzip.decompress(gz) == a + b


def test_damage_is_reported():
    data = TEXTS["reads"]
    gz = _gz(data)
    with pytest.raises(native.MercatHipError):
        native.gunzip(gz[:-9], len(data))                      # truncated
    with pytest.raises(native.MercatHipError):
        native.gunzip(gz[: len(gz) // 2], len(data))
    bad = bytearray(gz)
    bad[len(bad) // 2] ^= 0x55
    with pytest.raises(native.MercatHipError):
        native.gunzip(bytes(bad), len(data) + 100_000)          # corrupt data or CRC
    bad = bytearray(gz)
    bad[-6] ^= 1
    with pytest.raises(native.MercatHipError):
        native.gunzip(bytes(bad), len(data))                   # CRC field itself
    with pytest.raises(native.MercatHipError):
        native.gunzip(b">not gzip\nACGT\n", 100)
    with pytest.raises(native.MercatHipError):
        native.gunzip(gz + b"trailing garbage", len(data))


def test_fuzzed_streams_never_crash():
    rng = random.Random(3)
    data = TEXTS["reads"][:30_000]
    gz = bytearray(_gz(data))
    for _ in range(300):
        bad = bytearray(gz)
        for _ in range(rng.randint(1, 4)):
            bad[rng.randrange(10, len(bad))] = rng.randrange(256)
        try:
            out, _ = native.gunzip(bytes(bad), len(data) + 70_000, rng.choice([64, 4096, 1 << 20]))
            assert out == data        # (the damage hit bits that do not matter)
        except native.MercatHipError:
            pass


def test_crc32_equals_zlib():
    rng = random.Random(2)
    L = native.lib()
    blob = rng.randbytes(300_000)
    for _ in range(400):
        a = rng.randrange(0, 1000)
        n = rng.choice([0, 1, 15, 16, 63, 64, 65, 127, 128, 1000, rng.randrange(0, 200_000)])
        seed = rng.choice([0, rng.randrange(1 << 32)])
        piece = blob[a:a + n]
        addr, m, keep = native._buf_ptr(piece)
        assert L.mk_crc32_of(addr, m, seed) == zlib.crc32(piece, seed), (a, n, seed)


@pytest.mark.parametrize("name", ["reads", "text", "runs", "far", "random", "short", "empty"])
def test_parallel_decoder_equals_the_text(name):
    """Several threads on one DEFLATE stream (mk_pgunzip.h): block starts found by search, unknown history
    carried as place holders, pieces stitched together.  Small pieces put many seams into small files."""
    data = TEXTS[name] * (6 if name in ("reads", "text") else 1)
    for level, strategy in ((1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_FILTERED), (6, zlib.Z_FIXED), (0, 0)):
        gz = _gz(data, level, strategy)
        for threads, piece in ((1, 4096), (2, 4096), (4, 8192), (8, 4096), (3, 1 << 20)):
            got, members = native.gunzip_parallel(gz, len(data), threads, piece)
            assert got == data and members == 1, (name, level, strategy, threads, piece)


def test_parallel_decoder_members_and_damage():
    a, b = TEXTS["reads"] * 3, TEXTS["text"]
    gz = _gz(a, 6, memlevel=2) + b"\0" * 5 + _gz(b, 1) + gzip.compress(b"")
    got, members = native.gunzip_parallel(gz, len(a) + len(b), 4, 4096)
    assert got == a + b and members == 3
    one = _gz(a)
    with pytest.raises(native.MercatHipError):
        native.gunzip_parallel(one[:-5], len(a), 4, 4096)
    bad = bytearray(one)
    bad[len(bad) // 2] ^= 0x10
    with pytest.raises(native.MercatHipError):
        native.gunzip_parallel(bytes(bad), len(a) + 200_000, 4, 4096)
    rng = random.Random(4)
    for _ in range(60):       # damage anywhere: an error or (bits that do not matter) the text, never a crash
        bad = bytearray(one)
        for _ in range(rng.randint(1, 3)):
            bad[rng.randrange(10, len(bad))] = rng.randrange(256)
        try:
            assert native.gunzip_parallel(bytes(bad), len(a) + 500_000, rng.choice([2, 5]), 4096)[0] == a
        except native.MercatHipError:
            pass


@pytest.mark.parametrize("fail_round", [0, 1, 3])
def test_parallel_decoder_continues_front_to_back_when_a_round_gives_up(monkeypatch, fail_round):
    """Whatever makes a round of pieces give up (here: forced), decoding goes on front to back from the last
    verified block header, through member ends, and the next member is cut into pieces again."""
    monkeypatch.setenv("MK_PGUNZIP_FAIL_ROUND", str(fail_round))
    a, b = TEXTS["reads"] * 5, TEXTS["text"] + TEXTS["far"]
    gz = _gz(a, 6, memlevel=3) + _gz(b, 1)
    for threads, piece in ((3, 4096), (2, 65536)):
        got, members = native.gunzip_parallel(gz, len(a) + len(b), threads, piece)
        assert got == a + b and members == 2, (fail_round, threads, piece)
    bad = bytearray(gz)
    bad[len(gz) // 3] ^= 0x21
    with pytest.raises(native.MercatHipError):
        native.gunzip_parallel(bytes(bad), len(a) + len(b) + 300_000, 3, 4096)
