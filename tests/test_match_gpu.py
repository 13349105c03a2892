"""GPU parity of the MFMA matcher (through the C ABI) against the CPU checker: indices exact,
distances bit-identical."""
import numpy as np
import pytest

import object_detection_amd as oda
import oracle_lib as orc
from synth import sift_like_descriptors

pytestmark = pytest.mark.gpu


def _check(q, db, dim=None, matcher=None):
    m = matcher or oda.Matcher(max(len(q), 1), max(len(db), 1))
    idx_g, dist_g = m.top2(q, db, dim)
    idx_o, dist_o = orc.match_top2(q, db, dim)
    assert np.array_equal(idx_g, idx_o)
    assert np.array_equal(dist_g.view(np.uint32), dist_o.view(np.uint32))
    return idx_g, dist_g


@pytest.mark.parametrize("nq,nd", [(1, 1), (1, 2), (37, 211), (128, 32), (129, 33), (300, 5000), (1000, 1000)])
def test_top2_parity_sizes(nq, nd):
    q = sift_like_descriptors(nq, seed=nq)
    db = sift_like_descriptors(nd, seed=10_000 + nd)
    _check(q, db)


def test_ties_duplicates_and_missing():
    q = sift_like_descriptors(70, 1)
    db = sift_like_descriptors(400, 2)
    db[17] = q[3]
    db[40] = db[90] = db[333] = q[5]
    db[100:140] = db[100]                      # 40 identical rows
    q[9] = db[100]
    idx, dist = _check(q, db)
    assert idx[3, 0] == 17 and dist[3, 0] == 0
    assert list(idx[5]) == [40, 90]
    assert list(idx[9]) == [100, 101]
    idx, dist = _check(q, db[:1])
    assert np.all(idx[:, 1] == -1) and np.all(np.isinf(dist[:, 1]))
    idx, dist = _check(q, db[:0])
    assert np.all(idx == -1)


def test_shifted_rows_dim127_and_surf_dim64():
    a = sift_like_descriptors(150, 3)
    b = sift_like_descriptors(170, 4)
    qa, qb = oda.pack_sift_rows(a), oda.pack_sift_rows(b)
    i1, d1 = _check(qa, qb)
    i2, d2 = _check(a, b, dim=127)            # same rows without the shift: identical answer
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2)
    rng = np.random.RandomState(5)
    s1 = rng.standard_normal((90, 64)).astype(np.float32)
    s2 = rng.standard_normal((260, 64)).astype(np.float32)
    s1 /= np.linalg.norm(s1, axis=1, keepdims=True)
    s2 /= np.linalg.norm(s2, axis=1, keepdims=True)
    _check(s1, s2)


def test_good_match_filters():
    a = sift_like_descriptors(200, 6)
    b = sift_like_descriptors(300, 7)
    rng = np.random.RandomState(8)
    b[:120] = np.clip(a[:120] + rng.normal(0, 4, (120, 128)), 0, 255).astype(np.float32)
    m = oda.Matcher(200, 300)
    matches, dist, mn, mx = oda.match_sift_features(m, a, b)
    idx_o, dist_o = orc.match_top2(oda.pack_sift_rows(a), oda.pack_sift_rows(b))
    thr = max(2.5 * min(100.0, float(dist_o[:, 0].min())), 0.35)
    keep = dist_o[:, 0] <= thr
    assert np.array_equal(matches[:, 0], np.nonzero(keep)[0])
    assert np.array_equal(matches[:, 1], idx_o[keep, 0])
    assert len(matches) >= 100
    good, _ = oda.match_ratio_test(m, a, b)
    keep = dist_o_full = None
    idx_f, dist_f = orc.match_top2(a, b)
    keep = dist_f[:, 0] < np.float32(0.75) * dist_f[:, 1]
    assert np.array_equal(good[:, 0], np.nonzero(keep)[0]) and np.array_equal(good[:, 1], idx_f[keep, 0])


def test_large_parity_subset_and_property():
    """4096 x 65536 x 128 against the checker, then a 16384 x 131072 planted-neighbour property run."""
    db = sift_like_descriptors(65536, 11)
    rng = np.random.RandomState(12)
    perm = rng.permutation(65536)[:4096]
    q = np.clip(db[perm] + rng.normal(0, 8, (4096, 128)), 0, 255).astype(np.float32)
    m = oda.Matcher(16384, 131072)
    idx, dist = _check(q, db, matcher=m)
    assert (idx[:, 0] == perm).mean() > 0.99
    db2 = sift_like_descriptors(131072, 13)
    perm2 = rng.permutation(131072)[:16384]
    q2 = db2[perm2].copy()
    idx2, dist2 = m.top2(q2, db2)
    assert np.array_equal(idx2[:, 0], perm2.astype(np.int32))      # exact copies come back with distance 0
    assert np.all(dist2[:, 0] == 0) and np.all(dist2[:, 1] > 0)


def test_sharded_database_merge_equals_single_search():
    """Database split into 3 shards searched separately on the GPU, candidates merged on (score, index):
    identical to one search (what every rank computes after the all-gather in the multi-GPU path)."""
    import torch
    from object_detection_amd.distributed import ShardedDatabaseMatcher, merge_top2, shard_range
    db = sift_like_descriptors(5003, 21)
    q = sift_like_descriptors(700, 22)
    db[4000] = q[5]
    db[10] = db[4990] = q[9]
    m = oda.Matcher(700, 5003)
    ref_idx, ref_d, ref_s = m.top2(q, db, return_scores=True)
    o_idx, o_d, o_s = orc.match_top2(q, db, return_scores=True)
    assert np.array_equal(ref_idx, o_idx) and np.array_equal(ref_s.view(np.uint32), o_s.view(np.uint32))
    S, I, D = [], [], []
    for r in range(3):
        lo, hi = shard_range(len(db), r, 3)
        idx, d, s = m.top2(q, db[lo:hi], return_scores=True)
        S.append(torch.from_numpy(s)); D.append(torch.from_numpy(d))
        I.append(torch.from_numpy(np.where(idx >= 0, idx + lo, -1).astype(np.int32)))
    wi, wd = merge_top2(torch.stack(S), torch.stack(I), torch.stack(D))
    assert np.array_equal(wi.numpy(), ref_idx)
    assert np.array_equal(wd.numpy().view(np.uint32), ref_d.view(np.uint32))
    # the single-rank path of the wrapper (no process group) gives the same answer
    sm = ShardedDatabaseMatcher(len(db), 0, max_query=700)
    i1, d1 = sm.match(q, db)
    assert np.array_equal(i1, ref_idx) and np.array_equal(d1.view(np.uint32), ref_d.view(np.uint32))


def _many_pairs(P, cap, seed, m=None):
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(seed)
    nq = [int(v) for v in rng.randint(0, cap, P)]
    nd = [int(v) for v in rng.randint(0, cap, P)]
    nq[0], nd[1], nd[2], nq[3] = 0, 0, 1, 1
    Q = torch.zeros((P, cap, 128), dtype=torch.float32, device=dev)
    D = torch.zeros((P, cap, 128), dtype=torch.float32, device=dev)
    host = []
    for p in range(P):
        q = sift_like_descriptors(max(nq[p], 1), seed=seed * 1000 + p)[:nq[p]]
        d = sift_like_descriptors(max(nd[p], 1), seed=seed * 1000 + 500 + p)[:nd[p]]
        if nq[p] and nd[p] > 3:
            d[nd[p] // 2] = q[0]                     # an exact hit
        host.append((q, d))
        Q[p, :nq[p]] = torch.from_numpy(q).to(dev)
        D[p, :nd[p]] = torch.from_numpy(d).to(dev)
    cq = torch.tensor(nq, dtype=torch.int32, device=dev)
    cd = torch.tensor(nd, dtype=torch.int32, device=dev)
    idx = torch.full((P, cap, 2), -9, dtype=torch.int32, device=dev)
    dist = torch.zeros((P, cap, 2), dtype=torch.float32, device=dev)
    m = m or oda.Matcher(cap, cap, max_pairs=P)
    pairs = [dict(q=Q[p].data_ptr(), db=D[p].data_ptr(), nq=cq[p:].data_ptr(), nd=cd[p:].data_ptr(),
                  idx2=idx[p].data_ptr(), dist2=dist[p].data_ptr()) for p in range(P)]
    m.top2_batch_device(pairs, cap, cap, 128, 128, 0)
    torch.cuda.synchronize()
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    for p, (q, d) in enumerate(host):
        if nq[p] == 0:
            continue
        i_o, d_o = orc.match_top2(q, d)
        assert np.array_equal(idx[p, :nq[p]], i_o), p
        assert np.array_equal(dist[p, :nq[p]].view(np.uint32), d_o.view(np.uint32)), p
    return m


def test_many_pairs_with_device_counts_and_uneven_sizes():
    """40 pairs in one call with row counts in device memory, including empty and single-row
    sides; every pair equals the checker."""
    _many_pairs(40, 700, seed=11)


def test_more_pairs_than_one_launch_and_table_reuse():
    """300 pairs in one call (the pair table of a launch holds 256, so two launches) and six calls on
    one matcher with different tables (the device-side tables are used in rotation)."""
    m = _many_pairs(300, 96, seed=12)
    for call in range(5):
        _many_pairs(7 + 50 * call, 96, seed=20 + call, m=m)


def _batch_against_checker(host, cap, dim=128):
    """Runs all (q, d) pairs of `host` in one batched call with device-side row counts and compares
    every pair with the checker."""
    import torch
    dev = torch.device("cuda:0")
    P = len(host)
    Q = torch.zeros((P, cap, 128), dtype=torch.float32, device=dev)
    D = torch.zeros((P, cap, 128), dtype=torch.float32, device=dev)
    for p, (q, d) in enumerate(host):
        Q[p, :len(q)] = torch.from_numpy(q).to(dev)
        D[p, :len(d)] = torch.from_numpy(d).to(dev)
    cq = torch.tensor([len(q) for q, _ in host], dtype=torch.int32, device=dev)
    cd = torch.tensor([len(d) for _, d in host], dtype=torch.int32, device=dev)
    idx = torch.full((P, cap, 2), -9, dtype=torch.int32, device=dev)
    dist = torch.zeros((P, cap, 2), dtype=torch.float32, device=dev)
    m = oda.Matcher(cap, cap, max_pairs=P)
    pairs = [dict(q=Q[p].data_ptr(), db=D[p].data_ptr(), nq=cq[p:].data_ptr(), nd=cd[p:].data_ptr(),
                  idx2=idx[p].data_ptr(), dist2=dist[p].data_ptr()) for p in range(P)]
    m.top2_batch_device(pairs, cap, cap, dim, 128, 0)
    torch.cuda.synchronize()
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    for p, (q, d) in enumerate(host):
        i_o, d_o = orc.match_top2(np.ascontiguousarray(q[:, :dim]), np.ascontiguousarray(d[:, :dim]))
        assert np.array_equal(idx[p, :len(q)], i_o), p
        assert np.array_equal(dist[p, :len(q)].view(np.uint32), d_o.view(np.uint32)), p


def test_screened_batches_ties_scales_and_near_ties():
    """Batches go through the bf16 screening passes; the answer must still be the exact search's.
    Databases made of a few rows repeated many times (every query has hundreds of rows inside the
    screening margin: the candidate lists overflow and the query is ranked against all rows, ties go
    to the lowest index), rows that differ from a query by one unit in one element (far below bf16
    resolution), values scaled by 1e-3 and 1e3, signed unit-length rows of dimension 64, and
    dimension 127 of 128."""
    rng = np.random.RandomState(5)
    host = []
    base = sift_like_descriptors(600, seed=77)
    for p in range(12):
        q = sift_like_descriptors(300 + 17 * p, seed=200 + p)
        if p < 4:                                   # few distinct rows, many copies
            d = np.repeat(base[: 3 + p], 150, axis=0)[rng.permutation(150 * (3 + p))]
        elif p < 8:                                 # near-ties: copies of queries with one element nudged
            d = sift_like_descriptors(500, seed=300 + p)
            for t in range(0, 200):
                d[t] = q[t % len(q)]
                d[t, (7 * t) % 128] += 1.0 if t % 3 else 0.0
        else:
            d = sift_like_descriptors(450 + p, seed=400 + p)
        scale = np.float32([1.0, 1e-3, 1e3][p % 3])
        host.append((np.ascontiguousarray(q * scale), np.ascontiguousarray(d * scale)))
    _batch_against_checker(host, 960)
    _batch_against_checker(host, 960, dim=127)
    surf = []
    for p in range(10):
        q = rng.standard_normal((200 + p, 128)).astype(np.float32)
        d = rng.standard_normal((333 + p, 128)).astype(np.float32)
        q[:, 64:] = 0
        d[:, 64:] = 0
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        d[5] = q[3]
        surf.append((q, d))
    _batch_against_checker(surf, 400, dim=64)


def _adversarial_pair(nq, nd, e, seed):
    """Queries and rows placed on the half-way points of bf16 rounding so that the bf16 score of the
    true nearest row X is as much too large, and those of the runners-up A, B as much too small, as
    rounding to 8 significant bits allows: the query rounds down on the elements X lives on and up on
    the elements A and B live on. A screening margin below the true bound drops X (with 0.0085 |q| |d|
    in place of 0.016 |q| |d| every one of these queries loses its nearest row). `e` scales
    everything by a power of two (same rounding pattern at another exponent)."""
    rng = np.random.RandomState(seed)
    sc = np.float32(2.0 ** e)
    q = np.zeros((nq, 128), np.float32)
    d = (rng.uniform(0.0, 10.0, (nd, 128)) * sc).astype(np.float32)        # far away filler rows
    lo_v, hi_v = np.float32(128.49) * sc, np.float32(128.51) * sc         # round to 128 and 129 (x 2^e)
    want = np.zeros(nq, np.int64)
    first = rng.permutation(128)[:64]                                      # the elements X lives on
    second = np.setdiff1d(np.arange(128), first)
    for i in range(nq):
        # every query has its own random signs (rounding is symmetric in the sign), which puts the rows
        # built for the other queries far away from it
        sg = rng.choice(np.float32([-1.0, 1.0]), 128)
        q[i, first], q[i, second] = lo_v, hi_v
        q[i] *= sg
        x, a, b = [3 * i + int(v) for v in rng.permutation(3)]
        d[[x, a, b]] = 0
        d[x, first] = lo_v * sg[first]
        d[a, second] = hi_v * sg[second]
        d[b, second] = hi_v * sg[second]
        d[a, first[0]] = d[b, first[0]] = np.float32(-2.0) * sc * sg[first[0]]   # A, B slightly farther than X
        want[i] = x
    return q, d, want


def test_screening_margin_against_adversarial_rounding():
    """Rows on bf16 rounding boundaries (see _adversarial_pair): the screened search must still return
    the plain fp32 search's answer, which is the checker's."""
    import torch
    dev = torch.device("cuda:0")
    P, nq, nd = 10, 600, 2000
    cap_q, cap_d = 2000, 2000
    host = [_adversarial_pair(nq, nd, e, seed=900 + e) for e in (-9, -4, -1, 0, 0, 1, 2, 5, 8, 12)]
    Q = torch.zeros((P, cap_q, 128), dtype=torch.float32, device=dev)
    D = torch.zeros((P, cap_d, 128), dtype=torch.float32, device=dev)
    for p, (q, d, _) in enumerate(host):
        Q[p, :nq] = torch.from_numpy(q).to(dev)
        D[p, :nd] = torch.from_numpy(d).to(dev)
    res = []
    for screening in (True, False):
        idx = torch.full((P, cap_q, 2), -9, dtype=torch.int32, device=dev)
        dist = torch.zeros((P, cap_q, 2), dtype=torch.float32, device=dev)
        m = oda.Matcher(cap_q, cap_d, max_pairs=P)
        m.set_screening(screening)
        # row counts passed by value: with counts in device memory the call would assume an eighth of the
        # capacities and stay below the screening threshold (10 x 2000 x 2000 scores are above it); the
        # query rows past nq are zeros and their results are ignored
        pairs = [dict(q=Q[p].data_ptr(), db=D[p].data_ptr(), idx2=idx[p].data_ptr(), dist2=dist[p].data_ptr())
                 for p in range(P)]
        m.top2_batch_device(pairs, cap_q, nd, 128, 128, 0)
        torch.cuda.synchronize()
        res.append((idx.cpu().numpy()[:, :nq], dist.cpu().numpy()[:, :nq]))
    (i_s, d_s), (i_e, d_e) = res
    assert np.array_equal(i_s, i_e) and np.array_equal(d_s.view(np.uint32), d_e.view(np.uint32))
    for p, (q, d, want) in enumerate(host):
        i_o, d_o = orc.match_top2(q, d)
        assert np.array_equal(i_s[p], i_o), p
        assert np.array_equal(d_s[p].view(np.uint32), d_o.view(np.uint32)), p
        assert np.array_equal(i_o[:, 0], want), p       # the construction holds: X is the true nearest row


def test_million_row_database_through_64_screening_splits():
    """1024 queries against 1 048 576 rows (the row count of the 1 M x 1 M configuration): the database is
    cut into 64 splits of 16384 rows for the screening passes. Includes rows repeated 40 times inside one
    split (that split's candidate list overflows and the query goes to the overflow kernel; ties go to the
    lowest index),
    a pair of equal rows in different splits, and an exact copy as nearest row. Checked against the
    checker, and the plain fp32 search must return the same bits."""
    import torch
    dev = torch.device("cuda:0")
    nd, nq = 1 << 20, 1024
    base = sift_like_descriptors(65536, seed=7)
    db = np.concatenate([np.roll(base, r, axis=1) for r in range(nd // 65536)])
    rng = np.random.RandomState(17)
    perm = rng.permutation(nd)[:nq]
    q = np.clip(db[perm] + rng.normal(0, 8, (nq, 128)), 0, 255).astype(np.float32)
    db[500000:500040] = q[3]                    # 40 copies in one split: the list of that split overflows
    db[123] = db[900000] = q[5]                 # equal rows in the first and the last but one split
    db[1048575] = q[7]                          # the very last row
    idx_o, dist_o = orc.match_top2(q, db)
    assert list(idx_o[3]) == [500000, 500001] and list(idx_o[5]) == [123, 900000] and idx_o[7, 0] == 1048575
    tq = torch.from_numpy(q).to(dev)
    td = torch.from_numpy(db).to(dev)
    m = oda.Matcher(nq, nd)
    for screening in (True, False):
        m.set_screening(screening)
        idx = torch.full((nq, 2), -9, dtype=torch.int32, device=dev)
        dist = torch.zeros((nq, 2), dtype=torch.float32, device=dev)
        m.top2_device(tq.data_ptr(), nq, td.data_ptr(), nd, 128, 128, idx.data_ptr(), dist.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(idx.cpu().numpy(), idx_o), screening
        assert np.array_equal(dist.cpu().numpy().view(np.uint32), dist_o.view(np.uint32)), screening
        # two bf16 passes over the database (too few queries for a sampled first pass), or none
        assert m.last_screen_passes() == (2.0 if screening else 0.0)
    assert (idx_o[:, 0] == perm).mean() > 0.98


def test_production_dispatch_at_natural_size():
    """16384 queries x 262144 rows x 128 through the default switches: at this size the matcher by itself
    converts the database into its bf16 image (k_match_pack_rows), screens with k_match_screen4 and lets
    the first pass sample the rows -- the path behind the 1 M x 1 M figure -- which last_screen_passes() < 2
    shows. Compared with the CPU checker, bit for bit, including a row repeated 40 times inside one split
    (that split's candidate list overflows), equal rows in different splits (the lower index wins) and
    queries / rows placed on bf16 rounding boundaries (_adversarial_pair). The same search through the
    library's sharding (two shards of 131072 rows on this GPU, each large enough for the same path)."""
    import torch
    from object_detection_amd.distributed import shard_range
    dev = torch.device("cuda:0")
    nd, nq = 262144, 16384
    base = sift_like_descriptors(65536, seed=7)
    db = np.concatenate([np.roll(base, r, axis=1) for r in range(nd // 65536)])
    rng = np.random.RandomState(23)
    perm = rng.permutation(nd)[:nq]
    q = np.clip(db[perm] + rng.normal(0, 8, (nq, 128)), 0, 255).astype(np.float32)
    aq, ad, awant = _adversarial_pair(600, 2000, 0, seed=901)
    q[1000:1600] = aq
    db[140000:142000] = ad
    db[100000:100040] = q[3]                    # 40 copies inside one split
    db[123] = db[200000] = q[5]                 # equal rows in different splits (and different shards)
    db[nd - 1] = q[7]                           # the very last row
    idx_o, dist_o = orc.match_top2(q, db)
    assert list(idx_o[3]) == [100000, 100001] and list(idx_o[5]) == [123, 200000] and idx_o[7, 0] == nd - 1
    assert np.array_equal(idx_o[1000:1600, 0], 140000 + awant)      # the construction survives the embedding
    tq = torch.from_numpy(q).to(dev)
    td = torch.from_numpy(db).to(dev)
    m = oda.Matcher(nq, nd)
    idx = torch.full((nq, 2), -9, dtype=torch.int32, device=dev)
    dist = torch.zeros((nq, 2), dtype=torch.float32, device=dev)
    m.top2_device(tq.data_ptr(), nq, td.data_ptr(), nd, 128, 128, idx.data_ptr(), dist.data_ptr())
    torch.cuda.synchronize()
    assert 1.0 < m.last_screen_passes() < 2.0           # sampled first pass: the large-search path ran
    assert np.array_equal(idx.cpu().numpy(), idx_o)
    assert np.array_equal(dist.cpu().numpy().view(np.uint32), dist_o.view(np.uint32))
    m.close()

    sm = oda.ShardedMatcher.local([0, 0], max_query=nq, max_shard_rows=nd // 2)
    outs, ios = [], []
    for r in range(2):
        lo, hi = shard_range(nd, r, 2)
        i2 = torch.full((nq, 2), -7, dtype=torch.int32, device=dev)
        d2 = torch.full((nq, 2), -7.0, dtype=torch.float32, device=dev)
        outs.append((i2, d2))
        ios.append(dict(q=tq.data_ptr(), shard=td[lo:hi].data_ptr(), rows=hi - lo, offset=lo,
                        idx2=i2.data_ptr(), dist2=d2.data_ptr()))
    sm.search_all(ios, nq, 128, 128)
    sm.sync()
    assert 1.0 < sm.last_screen_passes() < 2.0
    for i2, d2 in outs:
        assert np.array_equal(i2.cpu().numpy(), idx_o)
        assert np.array_equal(d2.cpu().numpy().view(np.uint32), dist_o.view(np.uint32))
    sm.close()
