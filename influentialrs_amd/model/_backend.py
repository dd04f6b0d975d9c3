"""Glue between the nn.Module front-ends and the HIP engine: keeps one Engine per
network, (re)binds its parameters when they move or change, and offers the
row-level operations the task handlers need.  No arithmetic of the hot path is
done in torch here; torch supplies tensors, streams and collectives."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from .._lib import IRS_MASK_CAUSAL, IRS_MASK_IRN, IRS_SWEEP_BF16
from ..dist import ShardGroup
from ..engine import Comm, Engine, IrsError


class HipBackend:
    def __init__(self, net: nn.Module, mask_mode: int):
        self.net = net
        self.mask_mode = mask_mode
        self.engine: Optional[Engine] = None
        self._fp = None
        self.group: Optional[ShardGroup] = None
        self.sweep = IRS_SWEEP_BF16
        self.rank, self.world = 0, 1
        self._stale = False  # a training entry point ran since the derived weights were built
        self.comm: Optional[Comm] = None  # collectives below the C ABI (item-sharded search loops)

    def __deepcopy__(self, memo):
        """copy.deepcopy(net): the copy gets its own backend and builds its own engine on first use (an engine is a
        handle to device state bound to THIS module's parameter storage; it is never shared or copied)."""
        import copy
        new = HipBackend.__new__(HipBackend)
        memo[id(self)] = new
        new.net = copy.deepcopy(self.net, memo)
        new.mask_mode, new.sweep = self.mask_mode, self.sweep
        new.engine, new._fp, new.group, new._stale, new.comm = None, None, None, False, None
        new.rank, new.world = self.rank, self.world
        return new

    def set_sharding(self, rank: int, world: int, drop_full: bool = True):
        """Item-dimension sharding over `world` ranks (torch.distributed must be
        initialised by the caller when world > 1).  drop_full: keep only the shard's rows of
        project.weight / project.bias in the module (the engine binds local rows either way)."""
        if (rank, world) != (self.rank, self.world):
            self.rank, self.world = rank, world
            self.engine = None
        proj = getattr(self.net, "project", None)
        if drop_full and world > 1 and proj is not None and proj.weight.shape[0] == self.net.n_item:
            from ..engine import shard_bounds
            lo, hi = shard_bounds(self.net.n_item, world, rank)
            with torch.no_grad():
                # shrink the SAME Parameter objects: optimizers built earlier (InfluentialNet.__init__, IRSNN, Evaluator)
                # keep pointing at live tensors, and no reference to the full [n_item, d] storage survives
                for p_ in (proj.weight, proj.bias):
                    p_.data = p_.data[lo:hi].clone()
                    p_.grad = None
                proj.out_features = hi - lo
            self._fp = None

    def _fingerprint(self):
        return tuple((t.data_ptr(), t._version) for t in self.net.state_dict(keep_vars=True).values())

    def invalidate(self):
        """Weights were changed behind autograd's back (`.data` writes outside a training step, which bump no version
        counter): the next get() re-binds and re-derives (bf16 catalog, packed decoder streams, captured graphs)."""
        self._stale = True

    def get(self, n_seqs: int, n_rows: int, for_training: bool = False, may_update: bool = True) -> Engine:
        net = self.net
        dev = next(net.parameters()).device
        if dev.type != "cuda":
            raise IrsError("inference runs on the HIP engine only: move the network to a GPU "
                           "(there is no CPU fallback in this package)")
        need_new = (self.engine is None or self.engine.device != dev or self.engine.max_seqs < n_seqs
                    or self.engine.max_rows < n_rows)
        if need_new:
            prev_s = self.engine.max_seqs if self.engine else 0
            prev_r = self.engine.max_rows if self.engine else 0
            self.engine = None
            self.engine = Engine(
                n_item=net.n_item, n_user=getattr(net, "n_user", 0), d=net.embed_dim, max_len=net.max_len,
                n_heads=net.n_heads, ffn_dim=net.ffn_dim, n_layers=net.n_layers,
                u_dim=getattr(net, "u_embed_dim", 0) if self.mask_mode == IRS_MASK_IRN else 0,
                mask_mode=self.mask_mode, device=dev, max_rows=max(n_rows, prev_r, 64),
                max_seqs=max(n_seqs, prev_s, 8), max_k=100, rank=self.rank, world=self.world)
            self._fp = None
            self.group = ShardGroup(self.engine) if self.world > 1 else None
            if self.world > 1 and (self.comm is None or self.comm.device != dev):
                self.comm = Comm(dev)  # RCCL under torch.distributed's nccl backend, gloo callbacks otherwise
        fp = self._fingerprint()
        if for_training and self._fp is not None and [a for a, _ in fp] == [a for a, _ in self._fp]:
            # same storage: the CE entry points read project.* in place, nothing derived is used -- but whatever the
            # caller does with the gradients (an optimizer step, also through .data, which bumps no version counter)
            # leaves the bf16 catalog and the filter's norms behind: the next inference get() re-finalises.  A loss that
            # cannot reach an optimizer (no_grad / nothing requires grad: the Evaluator's eval-only loss, alternating with
            # rankings batch after batch) changes no weight and must not cost a re-pack of a multi-GB catalog every batch.
            self._stale = self._stale or may_update
            return self.engine
        if fp != self._fp or self._stale:
            sd = {k: v.detach() for k, v in net.state_dict(keep_vars=True).items() if v.dtype == torch.float32}
            self.engine.bind_state_dict(sd)
            self._fp = fp
            self._stale = for_training and may_update
        return self.engine

    # ---- row-level helpers (single shard or sharded group) ---------------
    def topk(self, xrows, k=100):
        if self.group is not None:
            return self.group.topk(xrows, k, self.sweep)
        return self.engine.score_topk(xrows, k, self.sweep)

    def gather(self, xrows, ids0):
        return self.group.gather(xrows, ids0) if self.group is not None else self.engine.score_gather(xrows, ids0)

    def count_before(self, xrows, ref_score, ref_id0, excl):
        if self.group is not None:
            return self.group.count_before(xrows, ref_score, ref_id0, excl)
        return self.engine.score_count_before(xrows, ref_score, ref_id0, excl)

    def lse(self, xrows):
        return self.group.lse(xrows) if self.group is not None else self.engine.score_lse(xrows)


class _ProjectCE(torch.autograd.Function):
    """mean over valid rows of CrossEntropy(project(x), label) -- nn.Linear + nn.CrossEntropyLoss of the reference's
    train_batch (influentialRS.py:278-310, evaluator.py:53-68) -- without the [M, n_item] logits: forward =
    irs_ce_forward (float32 log-sum-exp sweep + label gather), backward = irs_ce_grad_logits per row chunk (the
    softmax gradient written once by the sweep's epilogue) followed by the two plain GEMMs dX = G W, dW = G^T X."""
    ROWS = 8192          # rows per engine call
    CHUNK_BYTES = 1 << 30  # budget of the dL/dlogits chunk

    @staticmethod
    def forward(ctx, x, weight, bias, labels0, backend, may_update):
        M = x.shape[0]
        eng = backend.get(1, min(M, _ProjectCE.ROWS), for_training=True, may_update=may_update)
        xd = x.detach().contiguous()
        lse = torch.empty(M, dtype=torch.float32, device=x.device)
        tot = torch.zeros(3, dtype=torch.float64, device=x.device)
        for c0 in range(0, M, _ProjectCE.ROWS):
            c1 = min(M, c0 + _ProjectCE.ROWS)
            l, _, t = eng.ce_forward(xd[c0:c1], labels0[c0:c1])
            lse[c0:c1] = l
            tot += t
        _, n_valid, n_bad = tot.tolist()  # one host read per loss, as nn.CrossEntropyLoss's own target check costs
        if n_bad > 0:
            raise IndexError(f"Target out of bounds: {int(n_bad)} label(s) >= n_item {weight.shape[0]} (nn.CrossEntropyLoss "
                             "raises here, reference influentialRS.py:270,301)")
        ctx.save_for_backward(xd, weight, labels0, lse, tot)
        ctx.backend = backend
        ctx.n_valid = n_valid
        if n_valid == 0:  # every target is a pad: the reference's mean over an empty selection is nan, its gradient zero
            return torch.full((), float("nan"), dtype=torch.float32, device=x.device)
        return (tot[0] / tot[1]).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        xd, weight, labels0, lse, tot = ctx.saved_tensors
        M, d = xd.shape
        N = weight.shape[0]
        if ctx.n_valid == 0:
            return torch.zeros_like(xd), torch.zeros_like(weight), torch.zeros(N, dtype=torch.float32, device=xd.device), None, None, None
        eng = ctx.backend.get(1, min(M, _ProjectCE.ROWS), for_training=True)
        mc = max(32, min(_ProjectCE.ROWS, M, (_ProjectCE.CHUNK_BYTES // (4 * N)) // 32 * 32))
        G = torch.empty((mc, N), dtype=torch.float32, device=xd.device)
        dx = torch.empty_like(xd)
        dW = torch.zeros_like(weight)
        db = torch.zeros(N, dtype=torch.float32, device=xd.device)
        for c0 in range(0, M, mc):
            c1 = min(M, c0 + mc)
            Gc = eng.ce_grad_logits(xd[c0:c1], labels0[c0:c1], lse[c0:c1], 1.0, G)[:c1 - c0]
            torch.mm(Gc, weight, out=dx[c0:c1])
            dW.addmm_(Gc.t(), xd[c0:c1])
            db += Gc.sum(0)
        sc = (g.double() / tot[1]).to(torch.float32)  # dL/dloss / n_valid, kept on the device
        return dx * sc, dW * sc, db * sc, None, None, None


def project_ce(x: torch.Tensor, project: nn.Linear, labels0: torch.Tensor, backend: "HipBackend") -> torch.Tensor:
    """Scalar loss; x [M, d] rows of the decoder, labels0 [M] 0-based with -1 = ignored."""
    # (grad mode is off inside Function.forward: whether this loss can lead to a weight update is decided here)
    may_update = torch.is_grad_enabled() and (x.requires_grad or project.weight.requires_grad or project.bias.requires_grad)
    return _ProjectCE.apply(x, project.weight, project.bias, labels0, backend, may_update)


def pad_ragged_ids(lists: Sequence, device, minus: int = 1) -> torch.Tensor:
    """List of 1-based id sequences -> int64 [B, n] of 0-based ids, -1 padded."""
    n = max([len(a) for a in lists] + [1])
    out = np.full((len(lists), n), -1, dtype=np.int64)
    for i, a in enumerate(lists):
        a = np.asarray(a.detach().cpu().numpy() if torch.is_tensor(a) else a, dtype=np.int64)
        out[i, :len(a)] = a - minus
    return torch.from_numpy(out).to(device)


def make_scheduler(optimizer):
    """ReduceLROnPlateau(factor=0.5, patience=4) as the reference builds it
    (influentialRS.py:92-95); its `verbose=True` no longer exists in current torch."""
    from torch.optim import lr_scheduler
    return lr_scheduler.ReduceLROnPlateau(optimizer, factor=0.5, patience=4)
