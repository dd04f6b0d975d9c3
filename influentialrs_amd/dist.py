"""Item-dimension sharding across the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Rank r
holds rows [lo_r, hi_r) of project.weight / project.bias.  Per scoring call:

    rows (data-parallel)  --all_gather-->  every rank has all rows
    local sweep over the rank's item shard  ->  per-shard top-k / counts / (max, sumexp) / scores
    per-shard top-k lists, packed to ONE 64-bit key per entry (include/irs_hip.h: irs_pack_topk),
    --all_to_all-->  every rank receives the world's lists of ITS OWN rows and merges them
    (`topk_own`; `topk` = the all-gather form that leaves every row's merged list on every rank)

The payloads are small (M*k*8 bytes per rank), i.e. latency bound; nothing here is
reduced at bandwidth scale.  The reference has no counterpart (it only knows
nn.DataParallel, pipeline.py:43-44); this is the MI355X-native replacement.

`scorer` is anything with the Engine scoring methods (score_topk, pack_topk, merge_topk_keys,
score_gather, score_count_before, score_lse); in the product it is an
influentialrs_amd.engine.Engine.  The CPU rehearsal tests (gloo, world 2) pass
an oracle-backed stand-in to exercise exactly this collective logic.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


class ShardGroup:
    def __init__(self, scorer, group: Optional["dist.ProcessGroup"] = None):
        self.scorer = scorer
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        # gloo (the CPU rehearsal backend) is only dependable on host tensors: with device tensors and several
        # ranks sharing one GPU its collectives were seen to hang now and then.  Its payloads go through the host.
        backend = dist.get_backend(group) if self.world > 1 else ""
        self._via_host = backend == "gloo"
        # all_to_all: nccl (= RCCL) and gloo both implement it.  Decided ONCE from the backend's name, never by
        # catching an exception from a collective: a rank-local failure there must surface as an error, not send
        # this rank into a different collective than its peers sit in.
        self._a2a = backend in ("nccl", "gloo")

    # -- plumbing ---------------------------------------------------------
    def _all_gather(self, t: torch.Tensor) -> torch.Tensor:
        """[...] -> [world, ...] (same shape on every rank)."""
        t = t.contiguous()
        if self._via_host and t.is_cuda:
            h = t.cpu()
            out = torch.empty((self.world,) + tuple(h.shape), dtype=h.dtype)
            dist.all_gather_into_tensor(out.view(-1), h.view(-1), group=self.group)
            return out.to(t.device)
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out.view(-1), t.view(-1), group=self.group)
        return out

    def _all_reduce(self, t: torch.Tensor, op) -> torch.Tensor:
        if self._via_host and t.is_cuda:
            h = t.cpu()
            dist.all_reduce(h, op=op, group=self.group)
            t.copy_(h)
            return t
        dist.all_reduce(t, op=op, group=self.group)
        return t

    def gather_rows(self, local_rows: torch.Tensor) -> torch.Tensor:
        """Rows decoded data-parallel -> all rows on every rank, rank-major."""
        if self.world == 1:
            return local_rows
        g = self._all_gather(local_rows)
        return g.view(-1, local_rows.shape[-1])

    def my_slice(self, rows_per_rank: int) -> slice:
        return slice(self.rank * rows_per_rank, (self.rank + 1) * rows_per_rank)

    def _all_to_all(self, t: torch.Tensor) -> torch.Tensor:
        """[world, ...] -> [world, ...]: slice j goes to rank j; slice i of the result came from rank i."""
        t = t.contiguous()
        src = t.cpu() if (self._via_host and t.is_cuda) else t
        if self._a2a:
            out = torch.empty_like(src)
            dist.all_to_all_single(out.view(-1), src.view(-1), group=self.group)
        else:  # a backend without all_to_all (neither nccl nor gloo): all-gather, keep the own column
            allt = torch.empty((self.world,) + tuple(src.shape), dtype=src.dtype, device=src.device)
            dist.all_gather_into_tensor(allt.view(-1), src.view(-1), group=self.group)
            out = allt[:, self.rank].contiguous()
        return out.to(t.device)

    # -- combines ---------------------------------------------------------
    def topk_own(self, xrows: torch.Tensor, k: int, sweep: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Global top-k of THIS rank's rows.  `xrows` = all rows, rank-major ([world * B, d], from gather_rows); every
        rank sweeps them over its item shard, and one all_to_all of packed keys hands each rank the world's lists
        of its own B rows.  Returns (val[B,k], ids0[B,k], status[B])."""
        val, ids, st = self.scorer.score_topk(xrows, k, sweep)
        if self.world == 1:
            return val, ids, st
        B = xrows.shape[0] // self.world
        keys = self.scorer.pack_topk(val, ids).view(self.world, B, k)
        mv, mi = self.scorer.merge_topk_keys(self._all_to_all(keys))
        return mv, mi, (mi[:, -1] < 0).to(torch.int32) * 4

    def topk_own_lse(self, xrows: torch.Tensor, k: int, sweep: int):
        """topk_own plus the rows' global (max, sum exp) -- a beam step's scoring: the shard is swept ONCE for both
        (Engine.score_topk_lse).  Returns (val[B,k], ids0[B,k], status[B], max[world*B], sumexp[world*B])."""
        if not hasattr(self.scorer, "score_topk_lse"):
            v, i, st = self.topk_own(xrows, k, sweep)
            return (v, i, st) + tuple(self.lse(xrows))
        val, ids, st, m, s = self.scorer.score_topk_lse(xrows, k, sweep)
        if self.world == 1:
            return val, ids, st, m, s
        B = xrows.shape[0] // self.world
        keys = self.scorer.pack_topk(val, ids).view(self.world, B, k)
        mv, mi = self.scorer.merge_topk_keys(self._all_to_all(keys))
        gm = m.clone()
        self._all_reduce(gm, dist.ReduceOp.MAX)
        s = s * torch.exp(m - gm)
        s = torch.where(torch.isfinite(m), s, torch.zeros_like(s))
        self._all_reduce(s, dist.ReduceOp.SUM)
        return mv, mi, (mi[:, -1] < 0).to(torch.int32) * 4, gm, s

    def topk(self, xrows: torch.Tensor, k: int, sweep: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Global top-k of ALL rows on every rank (one all_gather of packed keys)."""
        val, ids, st = self.scorer.score_topk(xrows, k, sweep)
        if self.world == 1:
            return val, ids, st
        mv, mi = self.scorer.merge_topk_keys(self._all_gather(self.scorer.pack_topk(val, ids)))
        # fallback bit is informational per shard; FEWER_THAN_K only if the merged list is short
        st = (mi[:, -1] < 0).to(torch.int32) * 4
        return mv, mi, st

    def gather(self, xrows: torch.Tensor, ids0: torch.Tensor) -> torch.Tensor:
        """Exact scores at global ids: each id lives in exactly one shard, the
        others report -inf, so an elementwise max assembles the row."""
        s = self.scorer.score_gather(xrows, ids0)
        if self.world == 1:
            return s
        return self._all_reduce(s, dist.ReduceOp.MAX)

    def count_before(self, xrows, ref_score, ref_id0, excl_ids0) -> torch.Tensor:
        c = self.scorer.score_count_before(xrows, ref_score, ref_id0, excl_ids0)
        if self.world > 1:
            self._all_reduce(c, dist.ReduceOp.SUM)
        return c

    def lse(self, xrows: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Global (max, sum exp(e - max)) from per-shard pairs."""
        m, s = self.scorer.score_lse(xrows)
        if self.world == 1:
            return m, s
        gm = m.clone()
        self._all_reduce(gm, dist.ReduceOp.MAX)
        s = s * torch.exp(m - gm)
        s = torch.where(torch.isfinite(m), s, torch.zeros_like(s))
        self._all_reduce(s, dist.ReduceOp.SUM)
        return gm, s
