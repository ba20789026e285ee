"""Two ranks (processes) sharing the one GPU of the test box: chunks dealt round-robin, every
rank filters its own chunks, then the key-range all-to-all merge (mercat2_amd.dist.merge_ranks).
The concatenation of the ranks' exports in rank order must equal the single-process table.
Backend gloo here (RCCL refuses two ranks on one device); the exchange logic is the same."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import read_input

pytestmark = pytest.mark.gpu
WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _data():
    from mercat2_amd import native
    reads = native.synth_reads(150_000, 31, 120_000, 150, 32).tobytes()
    odd = read_input("edge_lengths.fa")  # N / IUPAC windows -> by-reference rows cross ranks too
    # the all-T 32-mer (the packed table's free-slot mark, kept beside the table) in the FIRST chunk, which
    # rank 0 counts: after the exchange it stands in the middle of the rows the last rank receives
    return b">polyT\n" + b"T" * 200 + b"\n" + reads + odd


def _chunks(data):
    from mercat2_amd.chunker import chunk_offsets
    offs = chunk_offsets(data, 3_000_000)
    return list(zip(offs[:-1], offs[1:]))


def _worker(rank, port, k, c, out):
    import torch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from mercat2_amd import native
        from mercat2_amd.dist import merge_ranks
        data = _data()
        with native.Counter(k, native.ALPHABET_NT2, device=0) as ctx:
            for a, b in _chunks(data)[rank::WORLD]:
                ctx.count_chunk(memoryview(data)[a:b], c)
            merge_ranks(ctx, 2 * k, device=torch.device("cuda", 0))
            kmers, counts = ctx.export()
        out[rank] = (kmers, counts)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,c", [(31, 2), (21, 1), (32, 2)])
def test_two_ranks_equal_one(k, c):
    from mercat2_amd import native
    data = _data()
    with native.Counter(k, native.ALPHABET_NT2) as ctx:
        for a, b in _chunks(data):
            ctx.count_chunk(memoryview(data)[a:b], c)
        want_k, want_c = ctx.export()
    assert len(_chunks(data)) >= 4
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker, args=(port, k, c, out), nprocs=WORLD, join=True)
        parts = [out[r] for r in range(WORLD)]
    # packed rows: rank order == key order; by-reference rows all sit on rank 0, so merge by key
    keys = np.concatenate([p[0].reshape(-1, k).view("S%d" % k).reshape(-1) for p in parts])
    cnts = np.concatenate([p[1] for p in parts])
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(keys[order], want_k.view("S%d" % k).reshape(-1))
    assert np.array_equal(cnts[order], want_c)
    # ownership: every packed (ACGT-only) row of rank 1 is above every packed row of rank 0
    def packed(p):
        ks = p[0].reshape(-1, k)
        ok = np.isin(ks, np.frombuffer(b"ACGT", dtype=np.uint8)).all(axis=1)
        return ks[ok].view("S%d" % k).reshape(-1)
    p0, p1 = packed(parts[0]), packed(parts[1])
    assert p0.size and p1.size and np.sort(p0)[-1] < np.sort(p1)[0]
