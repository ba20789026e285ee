"""ctypes wrapper of oracle/kmer_oracle.c (the C restatement of the reference).  TEST
INFRASTRUCTURE ONLY -- see the header of kmer_oracle.c.  Build with `make -C oracle`."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = HERE / "_build" / "liboracle.so"
        if not so.exists():
            subprocess.check_call(["make", "-C", str(HERE)])
        _LIB = C.CDLL(str(so))
        _LIB.oracle_count.restype = C.c_int
        _LIB.oracle_count.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint64, C.POINTER(C.c_void_p),
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        _LIB.oracle_free.argtypes = [C.c_void_p]
    return _LIB


def count(data: bytes, k: int, min_count: int):
    """(kmers as (rows,k) uint8, counts uint64), rows in byte-wise sorted order."""
    L = lib()
    buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(0, dtype=np.uint8)
    pk, pc, rows = C.c_void_p(), C.c_void_p(), C.c_size_t()
    rc = L.oracle_count(buf.ctypes.data if len(data) else None, len(data), k, min_count, C.byref(pk), C.byref(pc), C.byref(rows))
    if rc:
        raise RuntimeError("oracle_count failed: %d" % rc)
    n = rows.value
    kmers = np.ctypeslib.as_array(C.cast(pk, C.POINTER(C.c_uint8)), shape=(n * k,)).copy().reshape(n, k) if n else np.zeros((0, k), np.uint8)
    counts = np.ctypeslib.as_array(C.cast(pc, C.POINTER(C.c_uint64)), shape=(n,)).copy() if n else np.zeros(0, np.uint64)
    L.oracle_free(pk)
    L.oracle_free(pc)
    return kmers, counts


def count_dict(data: bytes, k: int, min_count: int) -> dict:
    kmers, counts = count(data, k, min_count)
    flat = kmers.tobytes().decode("ascii")
    return dict(zip((flat[i:i + k] for i in range(0, len(flat), k)), counts.tolist()))
