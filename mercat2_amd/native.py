"""ctypes binding of libmercat_hip.so (C ABI: include/mercat_hip.h).

The library is looked up next to this file (it is built in-tree by
``__graft_entry__.build()`` / ``make -C mercat2_amd/csrc``).  A missing library or a missing
HIP device raises -- nothing here or above falls back to a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional, Sequence, Tuple

import numpy as np

ALPHABET_NT2, ALPHABET_AA5, ALPHABET_RAW = 0, 1, 2
MODE_NAMES = {0: "dense", 1: "hash64", 2: "hash128", 3: "byref"}

MK_OK = 0
ERR_NAMES = {-1: "MK_ERR_ARG", -2: "MK_ERR_HIP", -3: "MK_ERR_NOMEM", -4: "MK_ERR_STATE",
             -5: "MK_ERR_NON_ASCII", -6: "MK_ERR_IO", -7: "MK_ERR_RANGE", -8: "MK_ERR_UNSUPPORTED"}

# every symbol include/mercat_hip.h declares (tests check the library exports each of them)
ABI_SYMBOLS = [
    "mk_create", "mk_destroy", "mk_last_error", "mk_reset", "mk_set_canonical", "mk_chunk_begin", "mk_chunk_feed",
    "mk_chunk_feed_device", "mk_chunk_end", "mk_count_device", "mk_export_size", "mk_export",
    "mk_write_tsv", "mk_export_pairs_device", "mk_import_pairs_device", "mk_export_exotic",
    "mk_import_exotic", "mk_words_per_key", "mk_merge_from", "mk_set_profiling", "mk_get_stats", "mk_reset_stats",
    "mk_chunk_cuts", "mk_synth_reads", "mk_version", "mk_count_file", "mk_stream_cuts",
    "mk_merged_export", "mk_write_merged_tsv", "mk_trim", "mk_alpha_stats", "mk_gunzip", "mk_crc32_of", "mk_gunzip_parallel",
    "mk_filter_min", "mk_remove_n", "mk_free", "mk_write_merged_tsv_t", "mk_write_merged_tsv_as_reference",
    "mk_owner_bounds", "mk_plan_contexts", "mk_bucket_rows_device", "mk_import_rows_device", "mk_merge_devices",
    "mk_export_size_multi", "mk_export_multi", "mk_write_tsv_multi", "mk_record_cuts", "mk_sample_keys", "mk_dense_bins_device",
    "mk_device_count", "mk_reset_for", "mk_textwrap", "mk_set_clean", "mk_clean_stats", "mk_clean_runs",
    "mk_export_stats", "mk_share_table",
]
MK_ABI = 4  # the number mk_version() must announce: struct layouts and signatures of include/mercat_hip.h as bound below
MERGE_RANGES, MERGE_GATHER, MERGE_BALANCED, MERGE_RCCL = 0, 1, 2, 4


class MercatHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "MK_ERR"), code, message))
        self.code = code


class NonAsciiInput(MercatHipError, UnicodeDecodeError.__base__):  # ValueError family, like a decode error
    pass


class CleanUnsupported(MercatHipError):
    """Clean mode (Counter.set_clean): the text holds something whose rewrite by removeN the GPU does not reproduce;
    nothing was counted -- count the text mk_remove_n produces instead."""


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("raw_bytes", "symbols", "windows", "exotic_windows", "chunks",
                                           "survivors", "rows", "table_slots")] + \
               [("mode", C.c_int32), ("profiled", C.c_int32)] + \
               [(n, C.c_double) for n in ("ms_parse", "ms_pack", "ms_count", "ms_exotic", "ms_filter", "ms_export")] + \
               [(n, C.c_uint64) for n in ("n_parse", "n_pack", "n_count", "n_exotic", "n_filter", "n_export")] + \
               [("ms_part", C.c_double), ("n_part", C.c_uint64), ("records", C.c_uint64), ("distinct", C.c_uint64),
                ("part_retries", C.c_uint64), ("part_reused", C.c_uint64), ("fused_chunks", C.c_uint64), ("fuse_spilled", C.c_uint64), ("parse_retries", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class FileStats(C.Structure):
    """mk_file_stats_t (include/mercat_hip.h)."""
    _fields_ = ([(n, C.c_uint64) for n in ("disk_bytes", "text_bytes", "chunks")] +
                [(n, C.c_int32) for n in ("gz", "chunked", "members", "threads", "contexts", "devices", "split_pieces", "pad_")] +
                [(n, C.c_double) for n in ("s_wait_io", "s_wait_gpu", "s_total", "s_merge",
                                           "s_setup", "s_scan", "s_feed", "s_retire", "s_drain")])

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "pad_"}


class ExportStats(C.Structure):
    """mk_export_stats_t (include/mercat_hip.h)."""
    _fields_ = ([(n, C.c_uint64) for n in ("rows", "bytes")] +
                [(n, C.c_double) for n in ("s_sort", "s_d2h", "s_format", "s_write", "s_total")])

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class CleanGpu(C.Structure):
    """mk_clean_gpu_t (include/mercat_hip.h)."""
    _fields_ = [(n, C.c_uint64) for n in ("raw_bytes", "symbols", "gc_count", "n_bytes", "n_runs", "header_lines", "last_runs")]


class MergeStats(C.Structure):
    """mk_merge_stats_t (include/mercat_hip.h)."""
    _fields_ = ([(n, C.c_uint64) for n in ("rows_in", "rows_out", "rows_moved", "bytes_moved", "max_owned")] +
                [(n, C.c_int32) for n in ("contexts", "devices", "peer_direct", "rccl")] +
                [(n, C.c_double) for n in ("s_bucket", "s_copy", "s_import", "s_total")])

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "pad_"}


class CleanStats(C.Structure):
    """mk_clean_stats_t (include/mercat_hip.h)."""
    _fields_ = [(n, C.c_uint64) for n in ("gc_count", "total_length", "records", "split_records", "pieces", "n_runs")] + \
               [("unsupported_record", C.c_int64)]


class AlphaStats(C.Structure):
    """mk_alpha_t (include/mercat_hip.h)."""
    _fields_ = [("observed", C.c_uint64), ("total", C.c_uint64), ("freq", C.c_uint64 * 11),
                ("sum_sq", C.c_double), ("sum_clnc", C.c_double)]


_LIB: Optional[C.CDLL] = None


def library_path() -> Path:
    return Path(os.environ.get("MERCAT_HIP_LIB", Path(__file__).resolve().parent / "libmercat_hip.so"))


def lib() -> C.CDLL:
    """Load libmercat_hip.so once; raise if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not path.exists():
        raise MercatHipError(-2, "HIP extension %s is missing: build it with `python -c 'import __graft_entry__ as g; "
                                 "g.build()'` or `make -C mercat2_amd/csrc` (there is no CPU fallback)" % path)
    L = C.CDLL(str(path))
    vp, u8p, u64p, szp = C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)
    sig = {
        "mk_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
        "mk_destroy": (None, [vp]),
        "mk_last_error": (C.c_char_p, [vp]),
        "mk_reset": (C.c_int, [vp]),
        "mk_reset_for": (C.c_int, [vp, C.c_uint64]),
        "mk_set_canonical": (C.c_int, [vp, C.c_int]),
        "mk_set_clean": (C.c_int, [vp, C.c_int, C.c_int]),
        "mk_clean_stats": (C.c_int, [vp, C.POINTER(CleanGpu)]),
        "mk_clean_runs": (C.c_int, [vp, u64p, u64p, C.c_size_t, szp]),
        "mk_chunk_begin": (C.c_int, [vp]),
        "mk_chunk_feed": (C.c_int, [vp, u8p, C.c_size_t]),
        "mk_chunk_feed_device": (C.c_int, [vp, u8p, C.c_size_t]),
        "mk_chunk_end": (C.c_int, [vp, C.c_uint64]),
        "mk_count_device": (C.c_int, [vp, u8p, C.c_size_t, C.c_uint64]),
        "mk_export_size": (C.c_int, [vp, szp]),
        "mk_export": (C.c_int, [vp, u8p, u64p, C.c_size_t]),
        "mk_write_tsv": (C.c_int, [vp, C.c_char_p, C.c_char_p, szp]),
        "mk_export_stats": (C.c_int, [vp, C.POINTER(ExportStats)]),
        "mk_export_pairs_device": (C.c_int, [vp, u64p, u64p, C.c_size_t, szp]),
        "mk_import_pairs_device": (C.c_int, [vp, u64p, u64p, C.c_size_t]),
        "mk_export_exotic": (C.c_int, [vp, u8p, u64p, C.c_size_t, szp]),
        "mk_import_exotic": (C.c_int, [vp, u8p, u64p, C.c_size_t]),
        "mk_words_per_key": (C.c_int, [vp]),
        "mk_merge_from": (C.c_int, [vp, vp]),
        "mk_share_table": (C.c_int, [vp, vp]),
        "mk_set_profiling": (C.c_int, [vp, C.c_int]),
        "mk_get_stats": (C.c_int, [vp, C.POINTER(Stats)]),
        "mk_reset_stats": (C.c_int, [vp]),
        "mk_chunk_cuts": (C.c_int, [u8p, C.c_size_t, C.c_uint64, u64p, C.c_size_t, szp]),
        "mk_synth_reads": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32,
                                     C.c_uint64, u8p, C.c_size_t, szp]),
        "mk_version": (C.c_char_p, []),
        "mk_count_file": (C.c_int, [C.POINTER(vp), C.c_int, C.c_char_p, C.c_uint64, C.c_uint64, C.c_int,
                                    C.POINTER(FileStats)]),
        "mk_merged_export": (C.c_int, [C.POINTER(vp), C.c_int, u8p, u64p, C.c_size_t, szp]),
        "mk_write_merged_tsv": (C.c_int, [C.POINTER(vp), C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_char_p, szp]),
        "mk_write_merged_tsv_as_reference": (C.c_int, [C.POINTER(vp), C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_char_p, szp]),
        "mk_write_merged_tsv_t": (C.c_int, [C.POINTER(vp), C.c_int, C.POINTER(C.c_char_p), C.c_char_p, szp]),
        "mk_trim": (C.c_int, [vp]),
        "mk_filter_min": (C.c_int, [vp, C.c_uint64]),
        "mk_remove_n": (C.c_int, [u8p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), szp, C.POINTER(CleanStats)]),
        "mk_free": (None, [C.c_void_p]),
        "mk_textwrap": (C.c_int, [u8p, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p), szp]),
        "mk_alpha_stats": (C.c_int, [vp, C.POINTER(AlphaStats)]),
        "mk_gunzip_parallel": (C.c_int, [u8p, C.c_size_t, u8p, C.c_size_t, C.c_int, C.c_size_t, szp, C.POINTER(C.c_int)]),
        "mk_crc32_of": (C.c_uint32, [u8p, C.c_size_t, C.c_uint32]),
        "mk_gunzip": (C.c_int, [u8p, C.c_size_t, u8p, C.c_size_t, C.c_size_t, szp, C.POINTER(C.c_int)]),
        "mk_stream_cuts": (C.c_int, [u8p, C.c_size_t, C.c_uint64, C.c_size_t, u64p, C.c_size_t, szp]),
        "mk_record_cuts": (C.c_int, [u8p, C.c_size_t, C.c_uint64, C.c_size_t, u64p, C.c_size_t, szp]),
        "mk_owner_bounds": (C.c_int, [C.c_int, C.c_int, u64p]),
        "mk_plan_contexts": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_int)]),
        "mk_bucket_rows_device": (C.c_int, [vp, u64p, C.c_int, u64p, C.c_size_t, u64p]),
        "mk_import_rows_device": (C.c_int, [vp, u64p, C.c_size_t]),
        "mk_merge_devices": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(MergeStats)]),
        "mk_device_count": (C.c_int, []),
        "mk_sample_keys": (C.c_int, [vp, C.c_size_t, u64p, C.c_size_t, szp]),
        "mk_dense_bins_device": (C.c_int, [vp, u64p, C.c_size_t, C.c_int]),
        "mk_export_size_multi": (C.c_int, [C.POINTER(vp), C.c_int, szp]),
        "mk_export_multi": (C.c_int, [C.POINTER(vp), C.c_int, u8p, u64p, C.c_size_t]),
        "mk_write_tsv_multi": (C.c_int, [C.POINTER(vp), C.c_int, C.c_char_p, C.c_char_p, szp]),
    }
    L.mk_version.restype = C.c_char_p
    ver = (L.mk_version() or b"").decode()
    try:
        abi = int(ver.split()[1].split(".")[0])
    except (IndexError, ValueError):
        abi = -1
    if abi != MK_ABI:
        raise MercatHipError(-4, "%s announces '%s' but this binding is written for ABI %d: rebuild the library "
                                 "(make -C mercat2_amd/csrc) -- struct layouts would not match" % (path, ver, MK_ABI))
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _LIB = L
    return L


def _buf_ptr(data) -> Tuple[int, int, object]:
    """(address, nbytes, keepalive) of a bytes-like object without copying."""
    if isinstance(data, np.ndarray):
        a = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        return a.ctypes.data, a.nbytes, a
    mv = memoryview(data)
    if mv.nbytes == 0:
        return 0, 0, None
    a = np.frombuffer(mv, dtype=np.uint8)  # works for read-only buffers (bytes, mmap) too
    return a.__array_interface__["data"][0], a.nbytes, a


def _big_u8(n: int) -> np.ndarray:
    """Writable uint8 buffer from an anonymous mmap.  (np.empty madvises huge pages for large
    blocks, which makes the first touch of every page very slow in some sandboxes.)"""
    import mmap
    if n == 0:
        return np.empty(0, dtype=np.uint8)
    return np.frombuffer(mmap.mmap(-1, n), dtype=np.uint8)


# ------------------------------------------------------------------------------ host helpers
def chunk_cuts(text, chunksize: int) -> np.ndarray:
    """Offsets at which the reference Chunker would start chunks 1.. (lib/mercat2_Chunker.py:39-59)."""
    L = lib()
    addr, n, keep = _buf_ptr(text)
    need = C.c_size_t(0)
    cap = 64
    while True:
        cuts = np.empty(cap, dtype=np.uint64)
        rc = L.mk_chunk_cuts(addr, n, int(chunksize), cuts.ctypes.data, cap, C.byref(need))
        if rc == MK_OK:
            return cuts[: need.value].copy()
        if rc != -7:
            raise MercatHipError(rc, "mk_chunk_cuts")
        cap = need.value


def remove_n(text, toupper: bool = False) -> Tuple[bytes, dict]:
    """mk_remove_n: (cleaned FASTA text, mk_clean_stats_t fields).  IndexError for a record to be split whose header is
    empty (as the reference raises it), NonAsciiInput for a byte >= 0x80 in a sequence line."""
    L = lib()
    addr, n, keep = _buf_ptr(text)
    out, out_len, st = C.c_void_p(), C.c_size_t(0), CleanStats()
    rc = L.mk_remove_n(addr, n, 1 if toupper else 0, C.byref(out), C.byref(out_len), C.byref(st))
    if rc == -7:
        raise IndexError("list index out of range")  # header.split()[0] of an empty header (lib/mercat2_fasta.py:40-41)
    if rc == -5:
        raise NonAsciiInput(rc, "record %d holds sequence byte(s) >= 0x80 (non-ASCII sequence text is not supported)" % st.unsupported_record)
    if rc:
        raise MercatHipError(rc, "mk_remove_n")
    try:
        data = C.string_at(out, out_len.value) if out.value else b""
    finally:
        if out.value:
            L.mk_free(out)
    return data, {n_: int(getattr(st, n_)) for n_, _ in st._fields_}


def textwrap_lines(text, width: int = 80) -> list:
    """mk_textwrap: the lines the library's restatement of textwrap.wrap gives (bytes each)."""
    L = lib()
    addr, n, keep = _buf_ptr(text)
    out, out_len = C.c_void_p(), C.c_size_t(0)
    rc = L.mk_textwrap(addr, n, int(width), C.byref(out), C.byref(out_len))
    if rc:
        raise MercatHipError(rc, "mk_textwrap")
    try:
        data = C.string_at(out, out_len.value) if out.value else b""
    finally:
        if out.value:
            L.mk_free(out)
    return data.split(b"\n")[:-1] if data else []


def default_streams(k: int, alphabet: int) -> int:
    """Contexts that pay off per GPU: a second one fills the gaps between the kernels of a chunk when
    keys are one word (nucleotide k <= 32, protein k <= 12: measured +18 % on S2); with two-word keys
    (33..64-mers) the long LDS-bound count kernels of two contexts only get in each other's way
    (measured -30 % at k = 63), and the by-reference modes gain nothing."""
    one_word = (alphabet == ALPHABET_NT2 and k <= 32) or (alphabet == ALPHABET_AA5 and k <= 12)
    return 2 if one_word else 1


def gunzip(gz, cap: int, block: int = 4 << 20) -> Tuple[bytes, int]:
    """(text, members) of a gzip file held in memory, through the file reader's own decoder (mk_gunzip)."""
    L = lib()
    addr, n, keep = _buf_ptr(gz)
    out = np.empty(max(cap, 1), dtype=np.uint8)
    written, members = C.c_size_t(0), C.c_int(0)
    rc = L.mk_gunzip(addr, n, out.ctypes.data, cap, int(block), C.byref(written), C.byref(members))
    if rc:
        raise MercatHipError(rc, "mk_gunzip")
    return out[: written.value].tobytes(), members.value


def gunzip_parallel(gz, cap: int, threads: int = 4, piece: int = 1 << 20) -> Tuple[bytes, int]:
    """(text, members) through the parallel decoder (mk_gunzip_parallel)."""
    L = lib()
    addr, n, keep = _buf_ptr(gz)
    out = np.empty(max(cap, 1), dtype=np.uint8)
    written, members = C.c_size_t(0), C.c_int(0)
    rc = L.mk_gunzip_parallel(addr, n, out.ctypes.data, cap, int(threads), int(piece), C.byref(written), C.byref(members))
    if rc:
        raise MercatHipError(rc, "mk_gunzip_parallel")
    return out[: written.value].tobytes(), members.value


def stream_cuts(text, chunksize: int, block: int) -> np.ndarray:
    """chunk_cuts through the streaming scanner of mk_count_file, the text handed over ``block`` bytes
    at a time; raises if the scanner lost, repeated or misplaced a byte."""
    L = lib()
    addr, n, keep = _buf_ptr(text)
    need = C.c_size_t(0)
    cap = 64
    while True:
        cuts = np.empty(cap, dtype=np.uint64)
        rc = L.mk_stream_cuts(addr, n, int(chunksize), int(block), cuts.ctypes.data, cap, C.byref(need))
        if rc == MK_OK:
            return cuts[: need.value].copy()
        if rc != -7:
            raise MercatHipError(rc, "mk_stream_cuts")
        cap = need.value


def record_cuts(text, piece: int, block: int = 1 << 20) -> np.ndarray:
    """Where mk_count_file cuts one filter unit that it spreads over several GPUs (mk_record_cuts): pieces of at
    least ``piece`` bytes that end where a record starts."""
    L = lib()
    addr, n, keep = _buf_ptr(text)
    need = C.c_size_t(0)
    cap = 64
    while True:
        cuts = np.empty(cap, dtype=np.uint64)
        rc = L.mk_record_cuts(addr, n, int(piece), int(block), cuts.ctypes.data, cap, C.byref(need))
        if rc == MK_OK:
            return cuts[: need.value].copy()
        if rc != -7:
            raise MercatHipError(rc, "mk_record_cuts")
        cap = need.value


def device_count() -> int:
    """HIP devices this process sees (mk_device_count)."""
    return int(lib().mk_device_count())


def owner_bounds(key_bits: int, n: int) -> list:
    """mk_owner_bounds: first key of owner 1..n-1 when [0, 2^key_bits) is cut into n equal ranges."""
    out = np.zeros(max(1, n - 1), dtype=np.uint64)
    rc = lib().mk_owner_bounds(int(key_bits), int(n), out.ctypes.data)
    if rc:
        raise MercatHipError(rc, "mk_owner_bounds")
    return [int(x) for x in out[: n - 1]]


def plan_contexts(devices: Sequence[int], streams: int) -> list:
    """mk_plan_contexts: device of every context, in creation order, so that mk_count_file's "chunk i -> ctxs[i mod
    nctx]" means device devices[i mod ndev] with the chunks of a device taking turns on its streams."""
    nd = len(devices)
    arr = (C.c_int * nd)(*[int(d) for d in devices])
    out = (C.c_int * (nd * int(streams)))()
    rc = lib().mk_plan_contexts(arr, nd, int(streams), out)
    if rc:
        raise MercatHipError(rc, "mk_plan_contexts")
    return list(out)


def _ctx_array(ctxs: Sequence["Counter"]):
    return (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])


def merge_devices(ctxs: Sequence["Counter"], flags: int = MERGE_RANGES) -> dict:
    """mk_merge_devices: sum the running tables of contexts on several GPUs (or several on one) in this process.
    MERGE_RANGES: ctxs[i] ends up with key range i; MERGE_GATHER: everything in ctxs[0]."""
    st = MergeStats()
    rc = lib().mk_merge_devices(_ctx_array(ctxs), len(ctxs), int(flags), C.byref(st))
    if rc:
        ctxs[0]._check(rc)
    return st.as_dict()


def rows_multi(ctxs: Sequence["Counter"]) -> int:
    n = C.c_size_t(0)
    rc = lib().mk_export_size_multi(_ctx_array(ctxs), len(ctxs), C.byref(n))
    if rc:
        ctxs[0]._check(rc)
    return n.value


def export_multi(ctxs: Sequence["Counter"]) -> Tuple[np.ndarray, np.ndarray]:
    """mk_export_multi: the sorted table of contexts that hold ascending key ranges (after MERGE_RANGES)."""
    rows = rows_multi(ctxs)
    kmers = np.empty((rows, ctxs[0].k), dtype=np.uint8)
    counts = np.empty(rows, dtype=np.uint64)
    rc = lib().mk_export_multi(_ctx_array(ctxs), len(ctxs), kmers.ctypes.data, counts.ctypes.data, rows)
    if rc:
        ctxs[0]._check(rc)
    return kmers, counts


def write_tsv_multi(ctxs: Sequence["Counter"], path, basename: str) -> int:
    n = C.c_size_t(0)
    rc = lib().mk_write_tsv_multi(_ctx_array(ctxs), len(ctxs), os.fsencode(str(path)), basename.encode(), C.byref(n))
    if rc:
        ctxs[0]._check(rc)
    return n.value


def count_file(ctxs: Sequence["Counter"], path, chunk_bytes: int, min_count: int, threads: int = 0) -> dict:
    """mk_count_file: read (inflate) ``path``, apply the Chunker rule iff its on-disk size is >=
    chunk_bytes > 0, count every chunk with its own min_count filter on the contexts in turn (they may sit on
    several GPUs: plan_contexts) and leave the sum in ctxs[0].  Returns the mk_file_stats_t fields."""
    L = lib()
    arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    st = FileStats()
    rc = L.mk_count_file(arr, len(ctxs), os.fsencode(str(path)), int(chunk_bytes), int(min_count), int(threads),
                         C.byref(st))
    if rc:
        ctxs[0]._check(rc)
    return st.as_dict()


def merged_export(ctxs: Sequence["Counter"]) -> Tuple[np.ndarray, np.ndarray]:
    """(kmers (rows, k) uint8, matrix (rows, len(ctxs)) uint64): every k-mer of any sample in sorted
    order with its count per sample, 0 where absent (mk_merged_export)."""
    L = lib()
    arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    rows = C.c_size_t(0)
    ctxs[0]._check(L.mk_merged_export(arr, len(ctxs), None, None, 0, C.byref(rows)))
    kmers = np.empty((rows.value, ctxs[0].k), dtype=np.uint8)
    matrix = np.empty((rows.value, len(ctxs)), dtype=np.uint64)
    if rows.value:
        ctxs[0]._check(L.mk_merged_export(arr, len(ctxs), kmers.ctypes.data, matrix.ctypes.data, rows.value, C.byref(rows)))
    return kmers, matrix


def write_merged_tsv(ctxs: Sequence["Counter"], names: Sequence[str], path, first_column: str = "k-mer",
                     as_reference: bool = False) -> int:
    """mk_write_merged_tsv: the combined table of merge_tsv (lib/mercat2_report.py:98-156) from the tables -- the true
    union, or with ``as_reference`` the rows exactly as the reference's streaming loop writes them (see the header)."""
    L = lib()
    arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    cn = (C.c_char_p * len(names))(*[n.encode() for n in names])
    rows = C.c_size_t(0)
    fn = L.mk_write_merged_tsv_as_reference if as_reference else L.mk_write_merged_tsv
    ctxs[0]._check(fn(arr, len(ctxs), cn, first_column.encode(), os.fsencode(str(path)), C.byref(rows)))
    return rows.value


def write_merged_tsv_T(ctxs: Sequence["Counter"], names: Sequence[str], path) -> int:
    """mk_write_merged_tsv_t: the file merge_tsv_T (lib/mercat2_report.py:160-194) writes, columns sorted."""
    L = lib()
    arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    cn = (C.c_char_p * len(names))(*[n.encode() for n in names])
    rows = C.c_size_t(0)
    ctxs[0]._check(L.mk_write_merged_tsv_t(arr, len(ctxs), cn, os.fsencode(str(path)), C.byref(rows)))
    return rows.value


def synth_reads(genome_len: int, genome_seed: int, reads: int, read_len: int, read_seed: int,
                sub_ppm: int = 0, first_index: int = 0) -> np.ndarray:
    """Deterministic synthetic FASTA reads as a uint8 array."""
    L = lib()
    size = C.c_size_t(0)
    rc = L.mk_synth_reads(genome_len, genome_seed, reads, read_len, read_seed, sub_ppm, first_index, None, 0, C.byref(size))
    if rc:
        raise MercatHipError(rc, "mk_synth_reads(size)")
    out = _big_u8(size.value)
    rc = L.mk_synth_reads(genome_len, genome_seed, reads, read_len, read_seed, sub_ppm, first_index,
                          out.ctypes.data, out.nbytes, C.byref(size))
    if rc:
        raise MercatHipError(rc, "mk_synth_reads")
    return out


# ----------------------------------------------------------------------------------- context
class Counter:
    """One GPU counting context for a fixed (alphabet, k): wraps mk_ctx."""

    def __init__(self, k: int, alphabet: int = ALPHABET_NT2, device: int = 0, canonical: bool = False):
        self._L = lib()
        self._h = C.c_void_p()
        self.k, self.alphabet, self.device = int(k), int(alphabet), int(device)
        rc = self._L.mk_create(self.device, self.alphabet, self.k, C.byref(self._h))
        if rc:
            msg = self._L.mk_last_error(None)
            raise MercatHipError(rc, msg.decode() if msg else "mk_create")
        if canonical:
            self.set_canonical(True)

    # -- plumbing
    def _check(self, rc: int):
        if rc:
            msg = self._L.mk_last_error(self._h)
            text = msg.decode() if msg else ""
            raise (NonAsciiInput if rc == -5 else CleanUnsupported if rc == -8 else MercatHipError)(rc, text)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.mk_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- counting
    def set_canonical(self, on: bool):
        """Opt-in extension (not reference behaviour): count min(kmer, reverse complement)."""
        self._check(self._L.mk_set_canonical(self._h, 1 if on else 0))

    def set_clean(self, on: bool, toupper: bool = False):
        """Count RAW nucleotide FASTA as removeN would leave it (mk_set_clean): N runs cut records, text in front of the
        first header is dropped, -toupper applies after the cut.  One chunk per file; CleanUnsupported when the text
        holds something the GPU does not reproduce."""
        self._check(self._L.mk_set_clean(self._h, 1 if on else 0, 1 if toupper else 0))

    def clean_stats(self) -> dict:
        st = CleanGpu()
        self._check(self._L.mk_clean_stats(self._h, C.byref(st)))
        return {n: int(getattr(st, n)) for n, _ in st._fields_}

    def clean_runs(self) -> Tuple[np.ndarray, np.ndarray]:
        """The N runs of the last chunk as (starts, ends) in the parsed stream (mk_clean_runs)."""
        n = C.c_size_t(0)
        self._check(self._L.mk_clean_runs(self._h, None, None, 0, C.byref(n)))
        a, b = np.empty(n.value, dtype=np.uint64), np.empty(n.value, dtype=np.uint64)
        if n.value:
            self._check(self._L.mk_clean_runs(self._h, a.ctypes.data, b.ctypes.data, n.value, C.byref(n)))
        return a, b

    def reset(self, expect_rows: int = 0):
        """Forget the running table; with ``expect_rows`` also size it for about that many keys if that is less than
        it has now (mk_reset_for: an owner about to take in its 1/N of a merged table)."""
        if expect_rows:
            self._check(self._L.mk_reset_for(self._h, int(expect_rows)))
        else:
            self._check(self._L.mk_reset(self._h))

    def count_chunk(self, data, min_count: int):
        """One reference find_kmers call: count ``data`` (raw FASTA bytes), keep >= min_count,
        add the survivors into the running table."""
        addr, n, keep = _buf_ptr(data)
        self._check(self._L.mk_chunk_begin(self._h))
        try:
            if n:
                self._check(self._L.mk_chunk_feed(self._h, addr, n))
        except Exception:
            self._L.mk_chunk_end(self._h, 0)
            raise
        self._check(self._L.mk_chunk_end(self._h, int(min_count)))

    def count_device(self, ptr: int, nbytes: int, min_count: int):
        """Count FASTA bytes already resident in this GPU's memory (ptr = device address)."""
        self._check(self._L.mk_count_device(self._h, ptr, int(nbytes), int(min_count)))

    # -- results
    def rows(self) -> int:
        n = C.c_size_t(0)
        self._check(self._L.mk_export_size(self._h, C.byref(n)))
        return n.value

    def export(self) -> Tuple[np.ndarray, np.ndarray]:
        """(kmers as a (rows, k) uint8 array, counts uint64) in sorted order."""
        rows = self.rows()
        kmers = np.empty((rows, self.k), dtype=np.uint8)
        counts = np.empty(rows, dtype=np.uint64)
        self._check(self._L.mk_export(self._h, kmers.ctypes.data, counts.ctypes.data, rows))
        return kmers, counts

    def to_dict(self) -> dict:
        kmers, counts = self.export()
        if kmers.shape[0] == 0:
            return {}
        flat = kmers.tobytes().decode("ascii")
        k = self.k
        return dict(zip((flat[i:i + k] for i in range(0, len(flat), k)), counts.tolist()))

    def write_tsv(self, path, basename: str) -> int:
        n = C.c_size_t(0)
        self._check(self._L.mk_write_tsv(self._h, os.fsencode(str(path)), basename.encode(), C.byref(n)))
        return n.value

    def export_stats(self) -> dict:
        """Where the last export / write_tsv of this context spent its time (mk_export_stats)."""
        st = ExportStats()
        self._check(self._L.mk_export_stats(self._h, C.byref(st)))
        return st.as_dict()

    # -- multi-GPU plumbing (device pointers come from torch tensors)
    def export_pairs_device(self, keys_ptr: int, counts_ptr: int, cap: int) -> int:
        n = C.c_size_t(0)
        self._check(self._L.mk_export_pairs_device(self._h, keys_ptr, counts_ptr, cap, C.byref(n)))
        return n.value

    def import_pairs_device(self, keys_ptr: int, counts_ptr: int, rows: int):
        self._check(self._L.mk_import_pairs_device(self._h, keys_ptr, counts_ptr, rows))

    def bucket_rows_device(self, bounds: Sequence[int], rows_ptr: int, cap_rows: int) -> list:
        """mk_bucket_rows_device: the table's rows grouped by owner (len(bounds)+1 owners) as interleaved
        {key word(s), count} rows in the device buffer at rows_ptr; returns the rows per owner."""
        n = len(bounds) + 1
        b = np.array(list(bounds) + [0], dtype=np.uint64)
        counts = np.zeros(n, dtype=np.uint64)
        self._check(self._L.mk_bucket_rows_device(self._h, b.ctypes.data, n, rows_ptr, int(cap_rows), counts.ctypes.data))
        return [int(x) for x in counts]

    def sample_keys(self, stride: int, cap: int = 1 << 16) -> np.ndarray:
        """About one in ``stride`` rows: the first word of their keys (mk_sample_keys), uint64, in no order."""
        out = np.empty(cap, dtype=np.uint64)
        n = C.c_size_t(0)
        self._check(self._L.mk_sample_keys(self._h, int(max(1, stride)), out.ctypes.data, cap, C.byref(n)))
        return out[: n.value].copy()

    def dense_bins_device(self, bins_ptr: int, nbins: int, store: bool):
        """Dense mode: copy the bins out to / in from a device buffer (mk_dense_bins_device)."""
        self._check(self._L.mk_dense_bins_device(self._h, bins_ptr, int(nbins), 1 if store else 0))

    def import_rows_device(self, rows_ptr: int, rows: int):
        self._check(self._L.mk_import_rows_device(self._h, rows_ptr, int(rows)))

    def words_per_key(self) -> int:
        """64-bit words per packed key in export_pairs_device / import_pairs_device (mk_words_per_key)."""
        return int(self._L.mk_words_per_key(self._h))

    def alpha_stats(self) -> dict:
        """Moments of the count column, reduced on the GPU (mk_alpha_stats)."""
        a = AlphaStats()
        self._check(self._L.mk_alpha_stats(self._h, C.byref(a)))
        return {"observed": int(a.observed), "total": int(a.total), "freq": [int(x) for x in a.freq],
                "sum_sq": float(a.sum_sq), "sum_clnc": float(a.sum_clnc)}

    def trim(self):
        """Free the per-chunk working memory, keep the running table (mk_trim)."""
        self._check(self._L.mk_trim(self._h))

    def filter_min(self, min_count: int):
        """Drop rows whose count is below ``min_count`` (the filter of a one-chunk sample counted in pieces)."""
        self._check(self._L.mk_filter_min(self._h, int(min_count)))

    def merge_from(self, other: "Counter"):
        """Add every row of ``other`` (same GPU, alphabet, k) into this context, on the device."""
        self._check(self._L.mk_merge_from(self._h, other._h))

    def share_table(self, owner: Optional["Counter"]):
        """From now on this context's count kernels put the survivors of its chunks into ``owner``'s running table (same
        GPU, alphabet, k; mk_share_table); ``None``: back to its own.  Sum with ``owner.merge_from(self)`` as before."""
        self._check(self._L.mk_share_table(self._h, owner._h if owner is not None else None))

    def export_exotic(self) -> Tuple[np.ndarray, np.ndarray]:
        n = C.c_size_t(0)
        self._check(self._L.mk_export_exotic(self._h, None, None, 0, C.byref(n)))
        kmers = np.empty((n.value, self.k), dtype=np.uint8)
        counts = np.empty(n.value, dtype=np.uint64)
        if n.value:
            self._check(self._L.mk_export_exotic(self._h, kmers.ctypes.data, counts.ctypes.data, n.value, C.byref(n)))
        return kmers, counts

    def import_exotic(self, kmers: np.ndarray, counts: np.ndarray):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint8)
        counts = np.ascontiguousarray(counts, dtype=np.uint64)
        if counts.size:
            self._check(self._L.mk_import_exotic(self._h, kmers.ctypes.data, counts.ctypes.data, counts.size))

    # -- stats
    def set_profiling(self, on: bool):
        self._check(self._L.mk_set_profiling(self._h, 1 if on else 0))

    def stats(self) -> dict:
        s = Stats()
        self._check(self._L.mk_get_stats(self._h, C.byref(s)))
        d = s.as_dict()
        d["mode_name"] = MODE_NAMES.get(d["mode"], "?")
        return d

    def reset_stats(self):
        self._check(self._L.mk_reset_stats(self._h))
