"""not-gpu: the N > 1 path rehearsed on CPU ranks (gloo, world_size 2 and 3).
The product's ShardGroup (influentialrs_amd/dist.py) runs unchanged; the
per-shard scorer is an oracle-backed stand-in for the Engine (test
infrastructure), so this checks exactly the collective logic: row all-gather,
per-shard top-k exchange (packed 64-bit keys: all-gather and all_to_all forms) + merge, max/sum combines."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from influentialrs_amd.engine import shard_bounds

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleShardScorer:
    """CPU stand-in with the Engine's scoring surface for one item shard."""

    def __init__(self, W, b, rank, world):
        import sys
        sys.path.insert(0, REPO)
        from oracle import oracle_np as O
        self.O = O
        self.lo, self.hi = shard_bounds(W.shape[0], world, rank)
        self.W, self.b = W[self.lo:self.hi], b[self.lo:self.hi]

    def _scores(self, xrows):
        return [self.O.score_chain(x, self.W, self.b) for x in xrows.numpy()]

    def score_topk(self, xrows, k, sweep):
        M = xrows.shape[0]
        val = torch.full((M, k), float("-inf"))
        ids = torch.full((M, k), -1, dtype=torch.int64)
        for m, s in enumerate(self._scores(xrows)):
            v, i = self.O.topk(s, k, id_base=self.lo)
            val[m, :len(v)] = torch.from_numpy(v)
            ids[m, :len(i)] = torch.from_numpy(i)
        return val, ids, torch.zeros(M, dtype=torch.int32)

    # the exchange step's wire format (include/irs_hip.h: irs_pack_topk / irs_merge_topk_keys), restated in numpy
    @staticmethod
    def _fkey(v):
        u = np.ascontiguousarray(v, dtype=np.float32).view(np.uint32).astype(np.uint64)
        u = np.where(u == 0x80000000, 0, u)
        return np.where(u & 0x80000000, ~u & 0xFFFFFFFF, u | 0x80000000)

    def pack_topk(self, val, ids):
        v, i = val.numpy(), ids.numpy()
        key = (self._fkey(v) << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - i.astype(np.uint64) & np.uint64(0xFFFFFFFF))
        key = np.where(i >= 0, key, np.uint64(0))
        return torch.from_numpy(key.view(np.int64).reshape(v.shape))

    def merge_topk_keys(self, keys):
        W, M, k = keys.shape
        ku = keys.numpy().view(np.uint64)
        ov = torch.full((M, k), float("-inf"))
        oi = torch.full((M, k), -1, dtype=torch.int64)
        for m in range(M):
            srt = np.sort(ku[:, m].reshape(-1))[::-1][:k]
            srt = srt[srt != 0]
            f = (srt >> np.uint64(32)).astype(np.uint32)
            bits = np.where(f & 0x80000000, f & 0x7FFFFFFF, ~f)
            ov[m, :len(srt)] = torch.from_numpy(bits.astype(np.uint32).view(np.float32).copy())
            oi[m, :len(srt)] = torch.from_numpy((np.uint64(0xFFFFFFFF) - (srt & np.uint64(0xFFFFFFFF))).astype(np.int64))
        return ov, oi

    def score_gather(self, xrows, ids0):
        out = torch.full(ids0.shape, float("-inf"))
        for m, s in enumerate(self._scores(xrows)):
            for j, g in enumerate(ids0[m].tolist()):
                if self.lo <= g < self.hi:
                    out[m, j] = float(s[g - self.lo])
        return out

    def score_count_before(self, xrows, ref_score, ref_id0, excl):
        out = torch.zeros(xrows.shape[0], dtype=torch.int64)
        for m, s in enumerate(self._scores(xrows)):
            rs, rid = float(ref_score[m]), int(ref_id0[m])
            ex = set(int(e) for e in excl[m].tolist() if e >= 0) if excl is not None else set()
            gid = np.arange(self.lo, self.hi)
            before = (s > rs) | ((s == rs) & (gid < rid))
            before &= ~np.isin(gid, list(ex - {rid}))
            out[m] = int(before.sum())
        return out

    def score_lse(self, xrows):
        mx, sm = [], []
        for s in self._scores(xrows):
            m, se = self.O.max_sumexp(s)
            mx.append(m)
            sm.append(se)
        return torch.tensor(mx, dtype=torch.float32), torch.tensor(sm, dtype=torch.float32)


def _worker(rank, world, port, n_item, d, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from influentialrs_amd.dist import ShardGroup
        g = np.random.default_rng(0)
        W = ((g.random((n_item, d), dtype=np.float32) * 2 - 1) / np.sqrt(d)).astype(np.float32)
        b = (g.standard_normal(n_item) * 0.1).astype(np.float32)
        rows_per_rank = 3
        x_all = g.standard_normal((world * rows_per_rank, d)).astype(np.float32)
        mine = torch.from_numpy(x_all[rank * rows_per_rank:(rank + 1) * rows_per_rank])
        grp = ShardGroup(OracleShardScorer(W, b, rank, world))
        assert grp.world == world and grp.rank == rank
        rows = grp.gather_rows(mine)  # data-parallel rows -> all rows everywhere
        assert np.array_equal(rows.numpy(), x_all)
        k = 20
        val, ids, st = grp.topk(rows, k, 0)
        full = OracleShardScorer(W, b, 0, 1)
        fv, fi, _ = full.score_topk(rows, k, 0)
        assert torch.equal(ids, fi) and torch.equal(val, fv)
        assert torch.equal(ids[grp.my_slice(rows_per_rank)], fi[rank * rows_per_rank:(rank + 1) * rows_per_rank])
        # the exchange the path-search loops use: one all_to_all of packed keys, own rows only
        ov, oi, _ = grp.topk_own(rows, k, 0)
        sl = grp.my_slice(rows_per_rank)
        assert torch.equal(oi, fi[sl]) and torch.equal(ov, fv[sl])
        kk = 1200  # more entries than the shard holds on some ranks: (-inf, -1) tails survive the packing
        tv, ti, _ = grp.topk_own(rows, kk, 0)
        gv, gi, _ = full.score_topk(rows, kk, 0)
        assert torch.equal(ti, gi[sl]) and torch.equal(tv, gv[sl])
        probe = torch.from_numpy(g.integers(0, n_item, size=(rows.shape[0], 5)).astype(np.int64))
        assert torch.equal(grp.gather(rows, probe), full.score_gather(rows, probe))
        lab = probe[:, 0].contiguous()
        ref = grp.gather(rows, lab.view(-1, 1))[:, 0].contiguous()
        excl = torch.from_numpy(g.integers(-1, n_item, size=(rows.shape[0], 7)).astype(np.int64))
        assert torch.equal(grp.count_before(rows, ref, lab, excl), full.score_count_before(rows, ref, lab, excl))
        m, s = grp.lse(rows)
        fm, fs = full.score_lse(rows)
        assert torch.equal(m, fm) and torch.allclose(s, fs, rtol=1e-5)
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_group_equals_single_shard(oracle, world):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 1000, 16, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert sorted(ret.keys()) == list(range(world))
