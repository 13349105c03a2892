"""The library's own database sharding (mif_match_sharded_*, through the C ABI): shards searched
separately, winners exchanged, merged on every rank -- against one search over the whole database and
against the CPU checker, bit for bit. On a one-GPU box the multi-shard cases place all shards on GPU 0
(exchange by device copies); the RCCL exchange runs with the one rank a single GPU allows."""
import ctypes as C

import numpy as np
import pytest

import object_detection_amd as oda
from object_detection_amd import _lib
from object_detection_amd.distributed import shard_range
import oracle_lib as orc
from synth import sift_like_descriptors

pytestmark = pytest.mark.gpu


def _run(torch, sm, q, db, nshards, dim=None, streams=None):
    """Shards db over the handle's local ranks, searches, returns every rank's (idx, dist)."""
    dev = torch.device("cuda:0")
    ld = q.shape[1]
    dim = ld if dim is None else dim
    tq = torch.from_numpy(q).to(dev)
    keep, ios, outs = [], [], []
    for r in range(nshards):
        lo, hi = shard_range(len(db), r, nshards)
        td = torch.from_numpy(np.ascontiguousarray(db[lo:hi])).to(dev) if hi > lo else None
        idx = torch.full((len(q), 2), -7, dtype=torch.int32, device=dev)
        dist = torch.full((len(q), 2), -7.0, dtype=torch.float32, device=dev)
        keep.append(td)
        outs.append((idx, dist))
        ios.append(dict(q=tq.data_ptr(), shard=td.data_ptr() if td is not None else None, rows=hi - lo, offset=lo,
                        idx2=idx.data_ptr(), dist2=dist.data_ptr(),
                        stream=streams[r].cuda_stream if streams else None))
    torch.cuda.synchronize()
    sm.search_all(ios, len(q), dim, ld)
    sm.sync()
    torch.cuda.synchronize()
    return [(i.cpu().numpy(), d.cpu().numpy()) for i, d in outs]


def _same(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


@pytest.mark.parametrize("nshards", [2, 3, 5])
def test_shards_on_one_gpu_equal_single_search(nshards):
    import torch
    db = sift_like_descriptors(5003, 21)
    q = sift_like_descriptors(700, 22)
    db[4000] = q[5]
    db[10] = db[4990] = q[9]                    # a tie across shards: the lower global index wins
    db[2500] = db[2501] = q[11]                 # both winners in one shard
    want = orc.match_top2(q, db)
    sm = oda.ShardedMatcher.local([0] * nshards, max_query=700, max_shard_rows=2600)
    assert not sm.uses_rccl()
    got = _run(torch, sm, q, db, nshards)
    for r in range(nshards):
        assert _same(got[r], want), r
    assert list(want[0][9]) == [10, 4990] and list(want[0][11]) == [2500, 2501]
    # again on the same handle with fewer queries, dimension 127 of 128 and caller-owned streams
    want2 = orc.match_top2(q[:97], db, 127)
    streams = [torch.cuda.Stream() for _ in range(nshards)]
    got2 = _run(torch, sm, q[:97], db, nshards, dim=127, streams=streams)
    for r in range(nshards):
        assert _same(got2[r], want2), r
    sm.close()


def test_more_shards_than_rows_and_single_row_database():
    import torch
    q = sift_like_descriptors(40, 31)
    db = sift_like_descriptors(2, 32)
    sm = oda.ShardedMatcher.local([0, 0, 0], max_query=64, max_shard_rows=8)
    got = _run(torch, sm, q, db, 3)             # shards of 1, 1 and 0 rows
    want = orc.match_top2(q, db)
    assert all(_same(g, want) for g in got)
    got = _run(torch, sm, q, db[:1], 3)         # one row in all: the second neighbour is missing everywhere
    want = orc.match_top2(q, db[:1])
    assert all(_same(g, want) for g in got)
    assert np.all(got[0][0][:, 1] == -1) and np.all(np.isinf(got[0][1][:, 1]))
    sm.close()


def test_rccl_exchange_with_the_one_rank_this_box_has():
    """nranks = 1 over RCCL: communicator, all-gather and merge run for real; the answer is the matcher's."""
    import torch
    db = sift_like_descriptors(3000, 41)
    q = sift_like_descriptors(500, 42)
    db[77] = q[3]
    sm = oda.ShardedMatcher(devices=[0], first_rank=0, nranks=1, unique_id=oda.ShardedMatcher.unique_id(),
                            max_query=500, max_shard_rows=3000)
    assert sm.uses_rccl()
    for screening in (True, False):
        sm.set_screening(screening)
        got = _run(torch, sm, q, db, 1)
        assert _same(got[0], orc.match_top2(q, db))
    sm.close()
    sm = oda.ShardedMatcher.local([0], max_query=500, max_shard_rows=3000)      # id made by the library
    assert sm.uses_rccl()
    assert _same(_run(torch, sm, q, db, 1)[0], orc.match_top2(q, db))
    sm.close()


def test_sharded_argument_errors():
    lib = oda.load_library()
    h = C.c_void_p()
    devs = (C.c_int * 2)(0, 0)
    assert lib.mif_match_sharded_create(devs, 2, 0, 3, None, 10, 10, C.byref(h)) == _lib.MIF_ERR_INVALID_ARGUMENT
    assert b"share a device" in lib.mif_last_error()
    one = (C.c_int * 1)(0)
    assert lib.mif_match_sharded_create(one, 1, 0, 2, None, 10, 10, C.byref(h)) == _lib.MIF_ERR_INVALID_ARGUMENT
    assert lib.mif_match_sharded_create(one, 1, 2, 2, None, 10, 10, C.byref(h)) == _lib.MIF_ERR_INVALID_ARGUMENT
    assert lib.mif_match_sharded_create(one, 0, 0, 1, None, 10, 10, C.byref(h)) == _lib.MIF_ERR_INVALID_ARGUMENT
    sm = oda.ShardedMatcher.local([0, 0], max_query=16, max_shard_rows=16)
    io = (_lib.ShardIO * 2)()
    assert lib.mif_match_sharded_search(sm._h, io, 17, 128, 128) == _lib.MIF_ERR_CAPACITY
    assert lib.mif_match_sharded_search(sm._h, io, 4, 128, 128) == _lib.MIF_ERR_INVALID_ARGUMENT
    sm.close()
