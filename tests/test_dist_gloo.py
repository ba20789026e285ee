"""The N > 1 merge path on CPU: world_size 2 over gloo.  exchange_pairs re-partitions sorted
(key,count) rows by key range with all_to_all; the test checks ownership and that the union
of what the ranks hold afterwards is the input, summed per key."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mercat2_amd import dist as mkdist

WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _table(rank, key_bits):
    rng = np.random.default_rng(100 + rank)
    keys = np.unique(rng.integers(0, 1 << min(key_bits, 62), 5000, dtype=np.uint64))
    if key_bits == 64:  # exercise the sign bit and the all-ones key
        keys = np.unique(np.concatenate([keys, keys | np.uint64(1 << 63), np.array([0xFFFFFFFFFFFFFFFF], dtype=np.uint64)]))
    cnts = rng.integers(1, 1000, keys.size).astype(np.uint64)
    return keys, cnts


def _worker(rank, port, key_bits, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        keys, cnts = _table(rank, key_bits)
        k = torch.from_numpy(keys.view(np.int64).copy())
        c = torch.from_numpy(cnts.view(np.int64).copy())
        rk, rc = mkdist.exchange_pairs(k, c, key_bits)
        out[rank] = (rk.numpy().view(np.uint64).copy(), rc.numpy().view(np.uint64).copy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("key_bits", [42, 62, 64])
def test_exchange_pairs_world2(key_bits):
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker, args=(port, key_bits, out), nprocs=WORLD, join=True)
        got = {r: out[r] for r in range(WORLD)}
    bounds = mkdist.range_bounds(key_bits, WORLD)
    want = {}
    for r in range(WORLD):
        keys, cnts = _table(r, key_bits)
        for k, c in zip(keys.tolist(), cnts.tolist()):
            want[k] = want.get(k, 0) + c
    merged = {}
    for r in range(WORLD):
        keys, cnts = got[r]
        lo = 0 if r == 0 else bounds[r - 1]
        hi = (1 << key_bits) if r == WORLD - 1 else bounds[r]
        assert all(lo <= k < hi for k in keys.tolist()), "rank %d received a key outside its range" % r
        for k, c in zip(keys.tolist(), cnts.tolist()):
            merged[k] = merged.get(k, 0) + c
    assert merged == want


def test_split_points_single_process():
    keys = torch.tensor([0, 1, 5, (1 << 61), (1 << 62) - 1], dtype=torch.int64)
    pts = mkdist.split_points(keys, 62, 4).tolist()
    assert pts == [0, 3, 3, 4, 5]  # bounds at 2^60, 2^61, 3*2^60
