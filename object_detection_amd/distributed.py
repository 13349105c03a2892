"""Multi-GPU host logic: one process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI on
MI355X nodes, "gloo" for CPU rehearsals).

Two ways the path shards (the reference itself has no multi-device code beyond its CPU/GPU split):
  * frames / images are independent: `shard_range` gives every rank a contiguous block and nothing is
    exchanged on the data path;
  * a descriptor database is split row-wise over the ranks: every rank searches its shard for the two
    nearest rows of every query, the per-rank candidates (score, global index, distance: 24 B per query and
    rank) are all-gathered, and every rank merges them to the global two nearest neighbours.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous block [lo, hi) of n items for `rank`; the first n % world ranks get one item more."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def merge_top2(scores, idx, dists):
    """Merges per-shard candidates into the global two nearest neighbours.

    scores, idx, dists: tensors [G, nq, 2] (ranking score, GLOBAL database index or -1, Euclidean
    distance). The winners are the two smallest (score, index) pairs -- the order one search over the
    whole database uses -- and are then reported nearest-first by (distance, index), exactly like a
    single-device search. Returns idx [nq, 2] int32 and dists [nq, 2] float32."""
    G, nq, _ = scores.shape
    s = scores.permute(1, 0, 2).reshape(nq, 2 * G)
    i = idx.permute(1, 0, 2).reshape(nq, 2 * G).to(torch.int64)
    d = dists.permute(1, 0, 2).reshape(nq, 2 * G)
    big = torch.iinfo(torch.int64).max
    i_key = torch.where(i < 0, torch.full_like(i, big), i)
    s_key = torch.where(i < 0, torch.full_like(s, float("inf")), s)
    # lexicographic (score, index): stable sort by index first, then stable sort by score
    o1 = torch.argsort(i_key, dim=1, stable=True)
    s1 = torch.gather(s_key, 1, o1)
    o2 = torch.argsort(s1, dim=1, stable=True)
    order = torch.gather(o1, 1, o2)[:, :2]
    wi = torch.gather(i, 1, order)
    wd = torch.gather(d, 1, order)
    wd = torch.where(wi < 0, torch.full_like(wd, float("inf")), wd)
    # nearest first by exact distance, ties to the lower index
    swap = (wi[:, 1] >= 0) & ((wd[:, 1] < wd[:, 0]) | ((wd[:, 1] == wd[:, 0]) & (wi[:, 1] < wi[:, 0])))
    sel = torch.stack([swap.to(torch.int64), 1 - swap.to(torch.int64)], dim=1)
    wi = torch.gather(wi, 1, sel)
    wd = torch.gather(wd, 1, sel)
    return wi.to(torch.int32), wd


class ShardedDatabaseMatcher:
    """Brute-force top-2 matching against a database split row-wise over the ranks of a process group.

    `local_top2(q, db_shard) -> (idx [nq,2] int32 local or -1, dist [nq,2] f32, score [nq,2] f32)` searches
    one shard; the default is the GPU matcher of this package. Queries are replicated on all ranks."""

    def __init__(self, shard_rows, global_offset, local_top2=None, group=None, device=0, max_query=None, dim=None):
        self.shard_rows = shard_rows
        self.global_offset = global_offset
        self.group = group
        self._local = local_top2
        self._device = device
        self._matcher = None
        self._max_query = max_query
        self.dim = dim

    def _gpu_local(self, q, db):
        from .match import Matcher
        if self._matcher is None:
            self._matcher = Matcher(self._max_query or max(len(q), 1), max(self.shard_rows, 1), device=self._device)
        return self._matcher.top2(q, db, self.dim, return_scores=True)

    def match(self, q, db_shard):
        """q [nq, ld] and db_shard [rows, ld] are host float32 arrays; returns global idx and dist on every rank."""
        local = self._local or self._gpu_local
        idx, d, s = local(np.ascontiguousarray(q, np.float32), np.ascontiguousarray(db_shard, np.float32))
        gidx = np.where(idx >= 0, idx + self.global_offset, -1).astype(np.int32)
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        backend = dist.get_backend(self.group) if dist.is_initialized() else "gloo"
        dev = torch.device("cuda", self._device) if backend == "nccl" else torch.device("cpu")
        # one packed message per rank: [nq, 2, 3] = (score, index bits, distance)
        packed = torch.stack([torch.from_numpy(s), torch.from_numpy(gidx).view(torch.float32),
                              torch.from_numpy(d)], dim=2).to(dev)
        if world > 1:
            gathered = [torch.empty_like(packed) for _ in range(world)]
            dist.all_gather(gathered, packed, group=self.group)
            allp = torch.stack(gathered)
        else:
            allp = packed[None]
        scores = allp[..., 0]
        idxs = allp[..., 1].contiguous().view(torch.int32)
        dists = allp[..., 2]
        wi, wd = merge_top2(scores, idxs, dists)
        return wi.cpu().numpy(), wd.cpu().numpy()


def sharded_sequence(n_frames, process, group=None):
    """Frame-to-previous-frame processing of a sequence split over the ranks of a process group.

    Every rank takes one contiguous block of frames (`shard_range`); a block that does not start the
    sequence is extended by the frame before it, so that its first frame is matched against the same
    predecessor as in a single-device run, and that extra frame's record is dropped. `process(lo, hi)`
    runs frames [lo, hi) in order and returns one record per frame (on the GPU: a VideoPipeline fed with
    those frames). No collective runs on the data path; the records are gathered at the end, and every
    rank returns the full list in frame order."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(n_frames, rank, world)
    mine = []
    if hi > lo:
        start = lo - 1 if lo > 0 else lo
        recs = list(process(start, hi))
        if len(recs) != hi - start:
            raise ValueError("process returned %d records for %d frames" % (len(recs), hi - start))
        mine = recs[lo - start:]
    if world == 1:
        return mine
    parts = [None] * world
    dist.all_gather_object(parts, mine, group=group)
    return [r for part in parts for r in part]
