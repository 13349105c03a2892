"""Host-side mirror of the reference's matching step (common/kpmatch.h) over the C ABI."""
import ctypes as C

import numpy as np

from . import _lib


class Matcher:
    """Exact brute-force L2 two-nearest-neighbour search on the GPU (replaces the FLANN
    calls of common/kpmatch.h:174-176 and 284-286)."""

    def __init__(self, max_query, max_database, device=0, max_pairs=1):
        self._lib = _lib.load_library()
        h = C.c_void_p()
        _lib.check(self._lib.mif_match_create_batched(device, max_query, max_database, max_pairs, C.byref(h)))
        self._h = h
        self.max_query, self.max_database, self.device = max_query, max_database, device
        self.max_pairs = max_pairs

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mif_match_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def set_screening(self, enable=True):
        """False: plain fp32 search for every call (same results; for measurements)."""
        _lib.check(self._lib.mif_match_set_screening(self._h, 1 if enable else 0))

    def last_screen_passes(self):
        """bf16 contraction passes over the database made by the last call's screening (0, 2 or 1 + 1/n)."""
        v = C.c_double()
        _lib.check(self._lib.mif_match_last_screen_passes(self._h, C.byref(v)))
        return v.value

    def top2(self, q, db, dim=None, return_scores=False):
        """q [nq, ld], db [nd, ld] float32 host arrays; only the first `dim` columns take part.
        Returns idx [nq, 2] int32 (-1 = missing) and dist [nq, 2] float32 (inf = missing); with
        return_scores also the ranking scores [nq, 2] used to merge database shards."""
        q = np.ascontiguousarray(q, np.float32)
        db = np.ascontiguousarray(db, np.float32)
        if q.ndim != 2 or db.ndim != 2 or (len(db) and q.shape[1] != db.shape[1]):
            raise ValueError("q and db must be 2-d with the same row length")
        ld = q.shape[1]
        dim = ld if dim is None else dim
        nq, nd = len(q), len(db)
        idx = np.full((nq, 2), -1, np.int32)
        dist = np.full((nq, 2), np.inf, np.float32)
        score = np.full((nq, 2), np.inf, np.float32)
        if nq:
            _lib.check(self._lib.mif_match_top2_scores(self._h, q.ctypes.data, nq, db.ctypes.data if nd else None,
                                                       nd, dim, ld, idx.ctypes.data, dist.ctypes.data,
                                                       score.ctypes.data if return_scores else None))
        return (idx, dist, score) if return_scores else (idx, dist)

    def top2_device(self, d_q, nq, d_db, nd, dim, ld, d_idx2, d_dist2, d_nq=None, d_nd=None, stream=None):
        """All pointers are integer device addresses; see mif_match_top2_device."""
        _lib.check(self._lib.mif_match_top2_device(
            self._h, C.c_void_p(d_q), nq, C.c_void_p(d_nq) if d_nq else None, C.c_void_p(d_db), nd,
            C.c_void_p(d_nd) if d_nd else None, dim, ld, C.c_void_p(d_idx2), C.c_void_p(d_dist2),
            C.c_void_p(stream) if stream else None))


    def top2_batch_device(self, pairs, nq, nd, dim, ld, filter_mode=0, stream=None):
        """pairs: list of dicts with integer device addresses q, db, idx2, dist2 and optionally
        nq, nd, good, summary, minmax (see mif_match_pair)."""
        arr = (_lib.MatchPair * len(pairs))()
        for a, p in zip(arr, pairs):
            a.d_q, a.d_db = p["q"], p["db"]
            a.d_nq, a.d_nd = p.get("nq"), p.get("nd")
            a.d_idx2, a.d_dist2 = p["idx2"], p["dist2"]
            a.d_good, a.d_summary, a.d_minmax = p.get("good"), p.get("summary"), p.get("minmax")
            a.d_score2 = p.get("score2")
        _lib.check(self._lib.mif_match_top2_batch_device(self._h, arr, len(pairs), nq, nd, dim, ld, filter_mode,
                                                         C.c_void_p(stream) if stream else None))


class ShardedMatcher:
    """A descriptor database sharded row-wise over several GPUs (mif_match_sharded_*): local top-2 search
    per shard, one RCCL all-gather of the winners, merge on every rank. Results equal one search over
    the whole database. All addresses are integer device addresses."""

    def __init__(self, devices, first_rank, nranks, unique_id, max_query, max_shard_rows):
        self._lib = _lib.load_library()
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        idbuf = C.create_string_buffer(bytes(unique_id), _lib.SHARD_ID_BYTES) if unique_id is not None else None
        _lib.check(self._lib.mif_match_sharded_create(devs, len(devices), first_rank, nranks, idbuf, max_query,
                                                      max_shard_rows, C.byref(h)))
        self._h = h
        self.devices, self.first_rank, self.nranks = list(devices), first_rank, nranks

    @staticmethod
    def unique_id():
        """The bytes one rank makes and hands to the others (RCCL's unique id)."""
        buf = C.create_string_buffer(_lib.SHARD_ID_BYTES)
        _lib.check(_lib.load_library().mif_match_sharded_unique_id(buf))
        return buf.raw

    @classmethod
    def local(cls, devices, max_query, max_shard_rows):
        """All ranks in this process, rank i on devices[i]."""
        return cls(devices, 0, len(devices), None, max_query, max_shard_rows)

    @classmethod
    def from_process_group(cls, dist, device, max_query, max_shard_rows):
        """One rank per process of an initialised torch.distributed group: rank 0 makes the id, the group
        broadcasts it, every rank joins the library's own RCCL communicator."""
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls([device], rank, world, box[0], max_query, max_shard_rows)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mif_match_sharded_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def ranks(self):
        """The handle's own view: {"nlocal", "nranks", "uses_rccl"} (mif_match_sharded_ranks)."""
        nl, nr, v = C.c_int(), C.c_int(), C.c_int()
        _lib.check(self._lib.mif_match_sharded_ranks(self._h, C.byref(nl), C.byref(nr), C.byref(v)))
        return {"nlocal": nl.value, "nranks": nr.value, "uses_rccl": bool(v.value)}

    def uses_rccl(self):
        v = C.c_int()
        _lib.check(self._lib.mif_match_sharded_ranks(self._h, None, None, C.byref(v)))
        return bool(v.value)

    def set_screening(self, enable=True):
        _lib.check(self._lib.mif_match_sharded_set_screening(self._h, 1 if enable else 0))

    def last_screen_passes(self):
        """bf16 contraction passes over the database made by the last search's screening."""
        v = C.c_double()
        _lib.check(self._lib.mif_match_sharded_last_screen_passes(self._h, C.byref(v)))
        return v.value

    def search_all(self, ios, nq, dim, ld):
        """ios: one dict per local rank with q, shard, rows, offset, idx2, dist2 and optionally stream."""
        arr = (_lib.ShardIO * len(ios))()
        for a, io in zip(arr, ios):
            a.d_q, a.d_shard = io["q"], io["shard"]
            a.shard_rows, a.global_offset = io["rows"], io["offset"]
            a.d_idx2, a.d_dist2 = io["idx2"], io["dist2"]
            a.stream = io.get("stream")
        _lib.check(self._lib.mif_match_sharded_search(self._h, arr, nq, dim, ld))

    def search(self, d_q, nq, d_shard, rows, offset, dim, ld, d_idx2, d_dist2, stream=None):
        """The call of a process that holds one rank."""
        self.search_all([dict(q=d_q, shard=d_shard, rows=rows, offset=offset, idx2=d_idx2, dist2=d_dist2,
                              stream=stream)], nq, dim, ld)

    def sync(self):
        _lib.check(self._lib.mif_match_sharded_sync(self._h))


def pack_sift_rows(descriptors):
    """Descriptor matrix as fmatches::ConvertToSift_feat + ConvertToCVKeyPoint build it
    (common/kpmatch.h:15,127-128,141-142): sift_feat::descriptor starts with one zero
    and the 128 values are appended, so matrix row = [0, d0 .. d126]."""
    d = np.asarray(descriptors, np.float32).reshape(-1, 128)
    out = np.zeros_like(d)
    out[:, 1:] = d[:, :127]
    return out


def match_sift_features(matcher, desc_object, desc_scene):
    """SIFT overload of Plot_Img_Matches_ocv up to the good-match list
    (common/kpmatch.h:174-202): nearest neighbour of every object row, then keep
    distance <= max(2.5 * min_dist, 0.35) with min_dist starting at 100.
    Returns (matches [M,2] int32 {queryIdx, trainIdx}, distances [M], min_dist, max_dist)."""
    q, db = pack_sift_rows(desc_object), pack_sift_rows(desc_scene)
    idx, dist = matcher.top2(q, db)
    d0 = dist[:, 0].astype(np.float64)
    min_dist, max_dist = 100.0, 0.0
    if len(d0):
        min_dist = min(min_dist, float(d0.min()))
        max_dist = max(max_dist, float(d0.max()))
    keep = d0 <= max(2.5 * min_dist, 0.35)          # in double, as kpmatch.h:198 compares
    qi = np.nonzero(keep)[0].astype(np.int32)
    return np.stack([qi, idx[keep, 0]], axis=1), dist[keep, 0], min_dist, max_dist


def match_ratio_test(matcher, desc_object, desc_scene, ratio=0.75):
    """SURF overload of Plot_Img_Matches_ocv up to the good-match list
    (common/kpmatch.h:284-297): 2-NN, keep d0 < 0.75 * d1."""
    idx, dist = matcher.top2(desc_object, desc_scene)
    keep = dist[:, 0] < np.float32(ratio) * dist[:, 1]
    qi = np.nonzero(keep)[0].astype(np.int32)
    return np.stack([qi, idx[keep, 0]], axis=1), dist[keep, 0]
