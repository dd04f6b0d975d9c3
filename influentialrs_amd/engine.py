"""Host-side engine: owns one irs_ctx (one device, one item shard), the derived
weight arena and the workspace (as torch tensors: torch is only the allocator
and stream provider here), and exposes the C-ABI calls on torch tensors.

All methods enqueue work on torch's current stream for the engine's device and
return device tensors; nothing synchronises unless the caller does.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import (IRS_MASK_CAUSAL, IRS_MASK_IRN, IRS_SWEEP_BF16, IRS_SWEEP_EXHAUSTIVE, IRS_SWEEP_F32, IrsDims,
                   IrsShard)


class IrsError(RuntimeError):
    pass


def shard_bounds(n_item: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous item shard [lo, hi) of rank `rank` out of `world`, sizes
    differing by at most one tile-aligned chunk: boundaries are multiples of 32
    items so that every shard but the last is made of whole MFMA tiles."""
    tiles = (n_item + 31) // 32
    per, rem = divmod(tiles, world)
    lo_t = rank * per + min(rank, rem)
    hi_t = lo_t + per + (1 if rank < rem else 0)
    return min(lo_t * 32, n_item), min(hi_t * 32, n_item)


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class Engine:
    def __init__(self, *, n_item: int, n_user: int, d: int, max_len: int, n_heads: int, ffn_dim: int, n_layers: int,
                 u_dim: int, mask_mode: int, device: torch.device, max_rows: int = 1024, max_seqs: int = 0,
                 max_k: int = 100, rank: int = 0, world: int = 1):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise IrsError("the HIP engine needs a GPU device (no CPU fallback exists)")
        lo, hi = shard_bounds(n_item, world, rank)
        if hi <= lo:
            raise IrsError(f"rank {rank} of {world} would hold an empty item shard of a {n_item}-item catalog")
        self.item_lo, self.item_hi = lo, hi
        self.n_local = hi - lo
        self.n_item, self.d, self.L = n_item, d, max_len
        self.max_rows, self.max_k = max_rows, max_k
        self.max_seqs = max_seqs or max_rows
        self.mask_mode = mask_mode
        self.world, self.rank = world, rank
        dims = IrsDims(n_item, n_user, d, max_len, n_heads, ffn_dim, n_layers, u_dim, mask_mode, max_rows, max_k,
                       self.max_seqs)
        shard = IrsShard(rank, world, lo, hi)
        h = ctypes.c_void_p()
        self._weights: Dict[str, torch.Tensor] = {}
        with torch.cuda.device(self.device):
            rc = self.lib.irs_create(ctypes.byref(h), ctypes.byref(dims), ctypes.byref(shard))
            if rc != 0:
                raise IrsError(f"irs_create failed ({rc}): {self.lib.irs_last_error(None).decode()}")
            self.h = h
            self._ws = torch.empty(self.lib.irs_workspace_bytes(self.h) + 256, dtype=torch.uint8, device=self.device)
            self._check(self.lib.irs_bind_workspace(self.h, _ptr(self._ws), self._ws.numel()))
        self._arena = None

    # ------------------------------------------------------------------
    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.irs_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise IrsError(f"libirs_hip error {rc}: {self.lib.irs_last_error(self.h).decode()}")

    def _call(self, fn, *args):
        """Every library entry point launches on the CURRENT HIP device (include/irs_hip.h): make the engine's
        device current around the call, whatever the caller's current device is."""
        with torch.cuda.device(self.device):
            self._check(fn(self.h, *args, self._stream()))

    def _inplace(self, t: torch.Tensor, dtype, what: str) -> torch.Tensor:
        """A buffer the library updates in place: it must already be what the kernels assume (a copy would
        silently drop the update)."""
        if t.device != self.device or t.dtype != dtype or not t.is_contiguous():
            raise IrsError(f"{what}: need a contiguous {dtype} tensor on {self.device}, got {t.dtype} on {t.device}"
                           f"{'' if t.is_contiguous() else ' (non-contiguous)'}")
        return t

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, t: torch.Tensor, dtype) -> torch.Tensor:
        if t.device != self.device:
            raise IrsError(f"tensor on {t.device}, engine on {self.device}")
        if t.dtype != dtype:
            raise IrsError(f"tensor dtype {t.dtype}, expected {dtype}")
        return t.contiguous()

    # ------------------------------------------------------------------ weights
    def bind_state_dict(self, sd: Dict[str, torch.Tensor]):
        """Bind every tensor of a reference-keyed state_dict.
        project.weight / project.bias may be given whole ([N, d] / [N]); the
        local shard rows are sliced here (a view, no copy)."""
        for name, t in sd.items():
            key = name[7:] if name.startswith("module.") else name
            if key.endswith("multihead_attn.in_proj_weight"):
                pass  # bound for the key-set contract; only its bias slice matters (zero memory)
            if key == "project.weight" and t.shape[0] == self.n_item and self.n_local != self.n_item:
                t = t[self.item_lo:self.item_hi]
            if key == "project.bias" and t.shape[0] == self.n_item and self.n_local != self.n_item:
                t = t[self.item_lo:self.item_hi]
            t = self._dev(t.detach(), torch.float32)
            self._weights[key] = t  # keep alive
            self._check(self.lib.irs_bind_weight(self.h, key.encode(), _ptr(t), t.numel()))
        self.finalize()

    def finalize(self):
        with torch.cuda.device(self.device):
            need = self.lib.irs_derived_bytes(self.h)
            if self._arena is None or self._arena.numel() < need:
                self._arena = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            self._check(self.lib.irs_finalize_weights(self.h, _ptr(self._arena), self._arena.numel(), self._stream()))

    # ------------------------------------------------------------------ decoder
    def pif(self, users: torch.Tensor) -> torch.Tensor:
        users = self._dev(users, torch.int64)
        out = torch.empty(users.shape[0], dtype=torch.float32, device=self.device)
        self._call(self.lib.irs_pif, _ptr(users), users.shape[0], _ptr(out))
        return out

    def decode(self, seqs: torch.Tensor, users: Optional[torch.Tensor], *, want_x: bool = True,
               pos: Optional[torch.Tensor] = None, want_r_u: bool = False):
        """Returns (x[B,L,d] or None, xrows[B,d] or None, r_u[B] or None)."""
        seqs = self._dev(seqs, torch.int64)
        B, L = seqs.shape
        if L != self.L:
            raise IrsError(f"sequence length {L} != max_len {self.L}")
        if users is not None:
            users = self._dev(users, torch.int64)
        x = torch.empty((B, L, self.d), dtype=torch.float32, device=self.device) if want_x else None
        xr = None
        if pos is not None:
            pos = self._dev(pos, torch.int32)
            xr = torch.empty((B, self.d), dtype=torch.float32, device=self.device)
        ru = torch.empty(B, dtype=torch.float32, device=self.device) if want_r_u else None
        self._call(self.lib.irs_decode, _ptr(seqs), _ptr(users), B, _ptr(x), _ptr(pos), _ptr(xr), _ptr(ru))
        return x, xr, ru

    # ------------------------------------------------------------------ scoring
    def score_topk(self, xrows: torch.Tensor, k: int = 100, sweep: int = IRS_SWEEP_BF16, carry: bool = False):
        """(val[M,k] float32, ids0[M,k] int64 global 0-based, status[M] int32) of this shard.  carry: the rows are the previous
        call's rows one path-search step later -- irs_score_topk_carry reuses that call's emission thresholds (same exact results;
        unrelated rows only cost time)."""
        xrows = self._dev(xrows, torch.float32)
        M = xrows.shape[0]
        val = torch.empty((M, k), dtype=torch.float32, device=self.device)
        ids = torch.empty((M, k), dtype=torch.int64, device=self.device)
        st = torch.empty(M, dtype=torch.int32, device=self.device)
        fn = self.lib.irs_score_topk_carry if (carry and sweep == IRS_SWEEP_BF16) else self.lib.irs_score_topk
        self._call(fn, _ptr(xrows), M, k, sweep, _ptr(val), _ptr(ids), _ptr(st))
        return val, ids, st

    def score_topk_lse(self, xrows: torch.Tensor, k: int = 100, sweep: int = IRS_SWEEP_BF16):
        """score_topk and score_lse out of one call (one pass over the float32 catalog on the swept path):
        (val, ids0, status, max[M], sumexp[M]) of this shard."""
        xrows = self._dev(xrows, torch.float32)
        M = xrows.shape[0]
        val = torch.empty((M, k), dtype=torch.float32, device=self.device)
        ids = torch.empty((M, k), dtype=torch.int64, device=self.device)
        st = torch.empty(M, dtype=torch.int32, device=self.device)
        mx = torch.empty(M, dtype=torch.float32, device=self.device)
        sm = torch.empty(M, dtype=torch.float32, device=self.device)
        self._call(self.lib.irs_score_topk_lse, _ptr(xrows), M, k, sweep, _ptr(val), _ptr(ids), _ptr(st), _ptr(mx), _ptr(sm))
        return val, ids, st, mx, sm

    def score_gather(self, xrows: torch.Tensor, ids0: torch.Tensor) -> torch.Tensor:
        xrows = self._dev(xrows, torch.float32)
        ids0 = self._dev(ids0, torch.int64)
        M, g = ids0.shape
        out = torch.empty((M, g), dtype=torch.float32, device=self.device)
        self._call(self.lib.irs_score_gather, _ptr(xrows), M, _ptr(ids0), g, _ptr(out))
        return out

    def score_count_before(self, xrows, ref_score, ref_id0, excl_ids0: Optional[torch.Tensor]) -> torch.Tensor:
        xrows = self._dev(xrows, torch.float32)
        ref_score = self._dev(ref_score, torch.float32)
        ref_id0 = self._dev(ref_id0, torch.int64)
        M = xrows.shape[0]
        n_ex = 0
        if excl_ids0 is not None:
            excl_ids0 = self._dev(excl_ids0, torch.int64)
            n_ex = excl_ids0.shape[1]
        out = torch.empty(M, dtype=torch.int64, device=self.device)
        self._call(self.lib.irs_score_count_before, _ptr(xrows), M, _ptr(ref_score), _ptr(ref_id0),
                                                    _ptr(excl_ids0) if n_ex else None, n_ex, _ptr(out))
        return out

    def score_dense(self, xrows: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        xrows = self._dev(xrows, torch.float32)
        M = xrows.shape[0]
        if out is None:
            out = torch.empty((M, self.n_local), dtype=torch.float32, device=self.device)
        self._call(self.lib.irs_score_dense, _ptr(xrows), M, _ptr(out), out.stride(0))
        return out

    def score_lse(self, xrows: torch.Tensor):
        xrows = self._dev(xrows, torch.float32)
        M = xrows.shape[0]
        mx = torch.empty(M, dtype=torch.float32, device=self.device)
        sm = torch.empty(M, dtype=torch.float32, device=self.device)
        self._call(self.lib.irs_score_lse, _ptr(xrows), M, _ptr(mx), _ptr(sm))
        return mx, sm

    # ---- projection + cross entropy without the logits (training side)
    def ce_forward(self, xrows: torch.Tensor, labels0: torch.Tensor):
        """(lse[M] float32, label_score[M] float32, loss[3] float64 = {sum over valid rows of lse - label score,
        number of valid rows, number of labels >= n_item}); labels0 int64 0-based, -1 = row ignored."""
        xrows = self._dev(xrows, torch.float32)
        labels0 = self._dev(labels0, torch.int64)
        M = xrows.shape[0]
        lse = torch.empty(M, dtype=torch.float32, device=self.device)
        ls = torch.empty(M, dtype=torch.float32, device=self.device)
        loss = torch.empty(3, dtype=torch.float64, device=self.device)
        self._call(self.lib.irs_ce_forward, _ptr(xrows), _ptr(labels0), M, _ptr(lse), _ptr(ls), _ptr(loss))
        return lse, ls, loss

    def ce_grad_logits(self, xrows: torch.Tensor, labels0: torch.Tensor, lse: torch.Tensor, scale: float, out: torch.Tensor):
        """out[M, ld >= n_local] = scale * (softmax(logits) - onehot(label)); ignored rows 0.  Written in place."""
        xrows = self._dev(xrows, torch.float32)
        labels0 = self._dev(labels0, torch.int64)
        lse = self._dev(lse, torch.float32)
        M = xrows.shape[0]
        if out.dtype != torch.float32 or out.device != self.device or out.dim() != 2 or out.stride(1) != 1 \
                or out.shape[0] < M or out.shape[1] < self.n_local:
            raise IrsError("ce_grad_logits: out must be a float32 [>= M, >= n_local] row-major tensor on the engine's device")
        self._call(self.lib.irs_ce_grad_logits, _ptr(xrows), _ptr(labels0), _ptr(lse), M, float(scale), _ptr(out),
                   int(out.stride(0)))
        return out

    def merge_topk(self, val_in: torch.Tensor, ids_in: torch.Tensor):
        """[W, M, k] gathered per-shard lists -> global (val[M,k], ids0[M,k])."""
        val_in = self._dev(val_in, torch.float32)
        ids_in = self._dev(ids_in, torch.int64)
        W, M, k = val_in.shape
        val = torch.empty((M, k), dtype=torch.float32, device=self.device)
        ids = torch.empty((M, k), dtype=torch.int64, device=self.device)
        self._call(self.lib.irs_merge_topk, _ptr(val_in), _ptr(ids_in), W, M, k, _ptr(val), _ptr(ids))
        return val, ids

    def pack_topk(self, val: torch.Tensor, ids0: torch.Tensor) -> torch.Tensor:
        """(val, ids0) lists -> the exchange step's 64-bit keys (include/irs_hip.h), same shape, int64 storage."""
        val = self._dev(val, torch.float32)
        ids0 = self._dev(ids0, torch.int64)
        keys = torch.empty(val.shape, dtype=torch.int64, device=self.device)
        self._call(self.lib.irs_pack_topk, _ptr(val), _ptr(ids0), val.numel(), _ptr(keys))
        return keys

    def merge_topk_keys(self, keys_in: torch.Tensor):
        """[W, M, k] packed per-shard lists -> global (val[M,k], ids0[M,k])."""
        keys_in = self._dev(keys_in, torch.int64)
        W, M, k = keys_in.shape
        val = torch.empty((M, k), dtype=torch.float32, device=self.device)
        ids = torch.empty((M, k), dtype=torch.int64, device=self.device)
        self._call(self.lib.irs_merge_topk_keys, _ptr(keys_in), W, M, k, _ptr(val), _ptr(ids))
        return val, ids

    # ------------------------------------------------------------------ evaluation batch (device-side loader)
    def build_eval_batch(self, items: torch.Tensor, offsets: torch.Tensor, *, raw_len: int = 100, gap_len: int = 0,
                         targets: Optional[torch.Tensor] = None, pool: Optional[torch.Tensor] = None, seed: int = 0):
        """get_random_evaluate_data + DataLoaderEvalIRS._collate_fn on the device (data_provider.py:398-449,
        591-617).  `items` int64 [total] / `offsets` int64 [B+1]: every user's events, oldest first.
        Returns (seq[B,L], targets[B], labels[B], raw[B,raw_len] right-aligned, raw_n[B] int32, status[B] int32)."""
        items = self._dev(items, torch.int64)
        offsets = self._dev(offsets, torch.int64)
        B = offsets.shape[0] - 1
        if targets is not None:
            targets = self._dev(targets, torch.int64)
        if pool is not None:
            pool = self._dev(pool, torch.int64)
        seq = torch.empty((B, self.L), dtype=torch.int64, device=self.device)
        tgt = torch.empty(B, dtype=torch.int64, device=self.device)
        lab = torch.empty(B, dtype=torch.int64, device=self.device)
        raw = torch.empty((B, raw_len), dtype=torch.int64, device=self.device)
        raw_n = torch.empty(B, dtype=torch.int32, device=self.device)
        status = torch.zeros(B, dtype=torch.int32, device=self.device)
        self._call(self.lib.irs_build_eval_batch, _ptr(items), _ptr(offsets), B, raw_len, gap_len, _ptr(targets),
                                                  _ptr(pool), 0 if pool is None else pool.shape[0], seed & (2 ** 64 - 1),
                                                  _ptr(seq), _ptr(tgt), _ptr(lab), _ptr(raw), _ptr(raw_n), _ptr(status))
        return seq, tgt, lab, raw, raw_n, status

    # ------------------------------------------------------------------ path search
    def path_step(self, seqs, hep, val, ids0, step: int, paths, status, sample=False, sample_k=3, seed=0):
        seqs = self._inplace(seqs, torch.int64, "path_step: seqs")
        hep = self._inplace(hep, torch.int32, "path_step: hep")
        paths = self._inplace(paths, torch.float32, "path_step: paths")
        status = self._inplace(status, torch.int32, "path_step: status")
        val, ids0 = self._dev(val, torch.float32), self._dev(ids0, torch.int64)
        B = seqs.shape[0]
        if seqs.shape[1] != self.L or hep.shape[0] != B or val.shape != ids0.shape or val.shape[0] != B:
            raise IrsError("path_step: inconsistent shapes")
        self._call(self.lib.irs_path_step, _ptr(seqs), _ptr(hep), B, _ptr(val), _ptr(ids0), val.shape[1], step,
                                           _ptr(paths), paths.shape[1], int(sample), sample_k, seed, _ptr(status))

    def generate_paths(self, seqs: torch.Tensor, users: Optional[torch.Tensor], hep: torch.Tensor, max_path_len: int,
                       k: int = 100, sweep: int = IRS_SWEEP_BF16, sample=False, sample_k=3, seed=0,
                       use_graph: bool = False, paths: Optional[torch.Tensor] = None,
                       status: Optional[torch.Tensor] = None):
        """Runs the whole search loop on the device.  `seqs` and `hep` are the
        working window state and are modified in place."""
        seqs = self._inplace(seqs, torch.int64, "generate_paths: seqs")
        hep = self._inplace(hep, torch.int32, "generate_paths: hep")
        if users is not None:
            users = self._dev(users, torch.int64)
        B = seqs.shape[0]
        if seqs.shape[1] != self.L or hep.shape[0] != B or (users is not None and users.shape[0] != B):
            raise IrsError("generate_paths: inconsistent shapes")
        if paths is None:
            paths = torch.zeros((B, max_path_len), dtype=torch.float32, device=self.device)
        if status is None:
            status = torch.zeros(B, dtype=torch.int32, device=self.device)
        paths = self._inplace(paths, torch.float32, "generate_paths: paths")
        status = self._inplace(status, torch.int32, "generate_paths: status")
        if paths.shape != (B, max_path_len):
            raise IrsError("generate_paths: paths must be [B, max_path_len]")
        self._call(self.lib.irs_generate_paths, _ptr(seqs), _ptr(users), _ptr(hep), B, max_path_len, k, sweep,
                                                int(sample), sample_k, seed, int(use_graph), _ptr(paths), _ptr(status))
        return paths, status

    # ------------------------------------------------------------------ beam search (build-defined extension)
    def beam_step(self, state_in, val, ids0, lse, step: int, state_out, status):
        """One beam step; state = (seq[B,W,L] i64, hep[B,W] i32, cum[B,W] f64, paths[B,W,P] f32)."""
        kinds = (torch.int64, torch.int32, torch.float64, torch.float32)
        seq_i, hep_i, cum_i, paths_i = (self._inplace(t, k, "beam_step: state_in") for t, k in zip(state_in, kinds))
        seq_o, hep_o, cum_o, paths_o = (self._inplace(t, k, "beam_step: state_out") for t, k in zip(state_out, kinds))
        status = self._inplace(status, torch.int32, "beam_step: status")
        val, ids0 = self._dev(val, torch.float32), self._dev(ids0, torch.int64)
        B, W, _ = seq_i.shape
        lmax, lsum = lse if lse is not None else (None, None)
        if lmax is not None:
            lmax, lsum = self._dev(lmax, torch.float32), self._dev(lsum, torch.float32)
        self._call(self.lib.irs_beam_step, _ptr(seq_i), _ptr(hep_i), _ptr(cum_i), _ptr(paths_i), _ptr(val),
                                           _ptr(ids0), _ptr(lmax), _ptr(lsum), B, W, val.shape[1], step,
                                           paths_i.shape[2], _ptr(seq_o), _ptr(hep_o), _ptr(cum_o), _ptr(paths_o),
                                           _ptr(status))

    def beam_search(self, seqs: torch.Tensor, users: Optional[torch.Tensor], hep: torch.Tensor, max_path_len: int,
                    beam: int, k: int = 100, sweep: int = IRS_SWEEP_BF16, use_graph: bool = False,
                    want_windows: bool = False):
        """(paths[B,W,P] f32, scores[B,W] f64, status[B] i32[, windows[B,W,L]]); beam 0 is the best."""
        seqs = self._dev(seqs, torch.int64)
        hep = self._dev(hep, torch.int32)
        if users is not None:
            users = self._dev(users, torch.int64)
        B = seqs.shape[0]
        paths = torch.zeros((B, beam, max_path_len), dtype=torch.float32, device=self.device)
        scores = torch.zeros((B, beam), dtype=torch.float64, device=self.device)
        status = torch.zeros(B, dtype=torch.int32, device=self.device)
        fin = torch.empty((B, beam, self.L), dtype=torch.int64, device=self.device) if want_windows else None
        self._call(self.lib.irs_beam_search, _ptr(seqs), _ptr(users), _ptr(hep), B, beam, max_path_len, k, sweep,
                                             int(use_graph), _ptr(paths), _ptr(scores), _ptr(fin), _ptr(status))
        return (paths, scores, status, fin) if want_windows else (paths, scores, status)

    # ------------------------------------------------------------------ item-sharded loops below the ABI (comm.hip)
    def allgather_rows(self, comm: "Comm", rows_local: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        rows_local = self._dev(rows_local, torch.float32)
        B = rows_local.shape[0]
        if out is None:
            out = torch.empty((comm.world * B, self.d), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.irs_allgather_rows(self.h, comm.h, _ptr(rows_local), B, _ptr(out), self._stream()))
        return out

    def exchange_topk(self, comm: "Comm", keys_send: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[world, B, k] packed lists of all rows (this shard) -> [world, B, k] the world's lists of this rank's rows."""
        keys_send = self._dev(keys_send, torch.int64)
        W, B, k = keys_send.shape
        if W != comm.world:
            raise IrsError("exchange_topk: keys must be [world, B, k]")
        if out is None:
            out = torch.empty_like(keys_send)
        with torch.cuda.device(self.device):
            self._check(self.lib.irs_exchange_topk(self.h, comm.h, _ptr(keys_send), _ptr(out), B, k, self._stream()))
        return out

    def generate_paths_sharded(self, comm: "Comm", seqs: torch.Tensor, users: Optional[torch.Tensor], hep: torch.Tensor,
                               max_path_len: int, k: int = 100, sweep: int = IRS_SWEEP_BF16, sample=False, sample_k=3, seed=0,
                               use_graph: bool = False, paths: Optional[torch.Tensor] = None,
                               status: Optional[torch.Tensor] = None):
        """generate_paths over the item-sharded catalog: this rank's B users; every rank calls it together."""
        seqs = self._inplace(seqs, torch.int64, "generate_paths_sharded: seqs")
        hep = self._inplace(hep, torch.int32, "generate_paths_sharded: hep")
        if users is not None:
            users = self._dev(users, torch.int64)
        B = seqs.shape[0]
        if seqs.shape[1] != self.L or hep.shape[0] != B or (users is not None and users.shape[0] != B):
            raise IrsError("generate_paths_sharded: inconsistent shapes")
        if paths is None:
            paths = torch.zeros((B, max_path_len), dtype=torch.float32, device=self.device)
        if status is None:
            status = torch.zeros(B, dtype=torch.int32, device=self.device)
        paths = self._inplace(paths, torch.float32, "generate_paths_sharded: paths")
        status = self._inplace(status, torch.int32, "generate_paths_sharded: status")
        if paths.shape != (B, max_path_len):
            raise IrsError("generate_paths_sharded: paths must be [B, max_path_len]")
        self._comm_ref = comm  # a captured step holds RCCL nodes of this communicator: it must outlive the context's graphs
        with torch.cuda.device(self.device):
            self._check(self.lib.irs_generate_paths_sharded(self.h, comm.h, _ptr(seqs), _ptr(users), _ptr(hep), B, max_path_len, k,
                                                            sweep, int(sample), sample_k, seed, int(use_graph), _ptr(paths),
                                                            _ptr(status), self._stream()))
        return paths, status

    def beam_search_sharded(self, comm: "Comm", seqs: torch.Tensor, users: Optional[torch.Tensor], hep: torch.Tensor,
                            max_path_len: int, beam: int, k: int = 100, sweep: int = IRS_SWEEP_BF16, split_decode: bool = False,
                            use_graph: bool = False, want_windows: bool = False):
        """beam_search over the item-sharded catalog.  split_decode=False: this rank's own B users; True: the SAME users on
        every rank, their B * beam windows decoded 1/world each (one user's beams spread over the node)."""
        seqs = self._dev(seqs, torch.int64)
        hep = self._dev(hep, torch.int32)
        if users is not None:
            users = self._dev(users, torch.int64)
        B = seqs.shape[0]
        paths = torch.zeros((B, beam, max_path_len), dtype=torch.float32, device=self.device)
        scores = torch.zeros((B, beam), dtype=torch.float64, device=self.device)
        # the status pointer is part of the captured step's reuse key (capi: sh_ptr): a buffer of the engine's own, so that
        # a replay does not depend on which address the allocator hands out; the caller receives a copy
        if getattr(self, "_beam_status", None) is None or self._beam_status.numel() < B:
            self._beam_status = torch.zeros(max(B, 64), dtype=torch.int32, device=self.device)
        status = self._beam_status[:B]
        status.zero_()
        fin = torch.empty((B, beam, self.L), dtype=torch.int64, device=self.device) if want_windows else None
        self._comm_ref = comm  # (see generate_paths_sharded)
        with torch.cuda.device(self.device):
            self._check(self.lib.irs_beam_search_sharded(self.h, comm.h, _ptr(seqs), _ptr(users), _ptr(hep), B, beam, max_path_len, k,
                                                         sweep, int(split_decode), int(use_graph), _ptr(paths), _ptr(scores),
                                                         _ptr(fin), _ptr(status), self._stream()))
        status = status.clone()
        return (paths, scores, status, fin) if want_windows else (paths, scores, status)

    # ------------------------------------------------------------------ decoder GEMM arithmetic
    @property
    def decoder_gemm(self) -> int:
        """IRS_GEMM_H3 (split-float16 MFMAs, the default), IRS_GEMM_X6 (split-bf16 MFMAs) or IRS_GEMM_F32 (float32 MFMAs):
        include/irs_hip.h.  The SELECTED mode (get / set round trips restore it); `decoder_gemm_effective` is what runs."""
        return int(self.lib.irs_get_decoder_gemm(self.h))

    @property
    def decoder_gemm_effective(self) -> int:
        """The arithmetic that runs: IRS_GEMM_X6 where IRS_GEMM_H3 is selected and the weights fail the float16 range bound."""
        return int(self.lib.irs_get_decoder_gemm_effective(self.h))

    @decoder_gemm.setter
    def decoder_gemm(self, mode: int):
        self._check(self.lib.irs_set_decoder_gemm(self.h, int(mode)))

    def debug_buffer(self, which: int, numel: int, dtype: torch.dtype) -> torch.Tensor:
        """(tests / lab) a view of one of the decoder's workspace buffers (irs_debug_ptr): what the last decode left behind."""
        p = self.lib.irs_debug_ptr(self.h, int(which))
        if not p:
            raise IrsError(f"irs_debug_ptr({which}) is null")
        off = int(p) - self._ws.data_ptr()
        nbytes = numel * torch.empty((), dtype=dtype).element_size()
        assert 0 <= off and off + nbytes <= self._ws.numel()
        return self._ws[off:off + nbytes].view(dtype)

    @property
    def decoder_seq(self) -> int:
        """Sequence-resident decoder (irs_set_decoder_seq, include/irs_hip.h): 0 off, 1 on, 2 auto (default: from 384 sequences
        per call up).  The setter takes False / True / None (= auto) or the numbers."""
        return int(self.lib.irs_get_decoder_seq(self.h))

    @decoder_seq.setter
    def decoder_seq(self, mode):
        self._check(self.lib.irs_set_decoder_seq(self.h, 2 if mode is None else int(mode)))

    @property
    def decoder_seq_last(self) -> bool:
        """True when the last decode took the sequence-resident path."""
        return bool(self.lib.irs_decoder_seq_last(self.h))

    @property
    def sharded_overlap(self) -> bool:
        """irs_set_sharded_overlap: generate_paths_sharded runs two user micro-batches per step with the collectives on a side
        stream (greedy choice; same results).  Off by default."""
        return bool(self.lib.irs_get_sharded_overlap(self.h))

    @sharded_overlap.setter
    def sharded_overlap(self, on: bool):
        self._check(self.lib.irs_set_sharded_overlap(self.h, 1 if on else 0))

    @property
    def h3_range_bound(self) -> float:
        """Largest operand magnitude the bound weights allow in the float16-plane kernels (irs_h3_range_bound; -1 before the
        weights are finalised).  At 32752 or more IRS_GEMM_H3 runs as IRS_GEMM_X6 and `decoder_gemm_effective` reports that."""
        return float(self.lib.irs_h3_range_bound(self.h))

    # ------------------------------------------------------------------ measurement
    def prof_enable(self, family: int):
        with torch.cuda.device(self.device):
            self._check(self.lib.irs_prof_enable(self.h, family))

    def prof_read(self):
        n = ctypes.c_int32()
        ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        with torch.cuda.device(self.device):
            self._check(self.lib.irs_prof_read(self.h, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)))
        return n.value, ms.value, fl.value, by.value


class Comm:
    """irs_comm handle (include/irs_hip.h, multi-GPU section): the collectives of the item-sharded loops, enqueued on the
    engine's stream BELOW the C ABI.  With torch.distributed's "nccl" backend it is an RCCL communicator of its own
    (ncclCommInitRank over a unique id that rank 0 draws and the process group broadcasts); with "gloo" (the CPU
    rehearsal backend: several ranks, possibly on ONE GPU) the library calls back into gloo through the host.  world == 1
    without a process group gives a one-rank RCCL communicator (the single-GPU anchor runs the same code path)."""

    def __init__(self, device: torch.device, group=None, backend: Optional[str] = None):
        import torch.distributed as dist
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.group = group
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if inited else 1
        self.rank = dist.get_rank(group) if inited else 0
        if backend is None:
            backend = dist.get_backend(group) if inited else "nccl"
        self.backend = backend
        h = ctypes.c_void_p()
        if backend == "nccl":
            buf = ctypes.create_string_buffer(_lib.IRS_COMM_ID_BYTES)
            if self.rank == 0:
                if self.lib.irs_comm_unique_id(buf) != 0:
                    raise IrsError(f"irs_comm_unique_id: {self.lib.irs_comm_last_error().decode()}")
            if self.world > 1:
                box = [bytes(buf.raw)]
                dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                buf = ctypes.create_string_buffer(box[0], _lib.IRS_COMM_ID_BYTES)
            with torch.cuda.device(self.device):
                rc = self.lib.irs_comm_init_rccl(ctypes.byref(h), buf, self.rank, self.world)
            if rc != 0:
                raise IrsError(f"irs_comm_init_rccl: {self.lib.irs_comm_last_error().decode()}")
        else:
            self._make_callbacks()
            rc = self.lib.irs_comm_init_callbacks(ctypes.byref(h), self.rank, self.world, None,
                                                  ctypes.cast(self._cb[0], ctypes.c_void_p), ctypes.cast(self._cb[1], ctypes.c_void_p),
                                                  ctypes.cast(self._cb[2], ctypes.c_void_p))
            if rc != 0:
                raise IrsError(f"irs_comm_init_callbacks: {self.lib.irs_comm_last_error().decode()}")
        self.h = h

    def __del__(self):
        try:
            import sys
            if sys.is_finalizing():
                return  # interpreter shutdown: the HIP runtime may already be going down under ncclCommDestroy (observed: a hang)
            if getattr(self, "h", None):
                self.lib.irs_comm_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def is_rccl(self) -> bool:
        return bool(self.lib.irs_comm_is_rccl(self.h))

    def _make_callbacks(self):
        """gloo through the host: wait for the stream, stage the payload in host memory, run the collective, copy back.
        (Not capturable, not fast: it exists so that the N > 1 code path runs where no second GPU does.)"""
        import numpy as np
        import torch.distributed as dist
        hip = ctypes.CDLL("libamdhip64.so")  # the copy torch has already loaded
        hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        D2H, H2D = 2, 1
        world, group, dev = self.world, self.group, self.device

        def fetch(ptr, n, stream):
            with torch.cuda.device(dev):
                if hip.hipStreamSynchronize(stream) != 0:
                    raise RuntimeError("hipStreamSynchronize")
                h = np.empty(n, dtype=np.uint8)
                if hip.hipMemcpy(h.ctypes.data, ptr, n, D2H) != 0:
                    raise RuntimeError("hipMemcpy D2H")
            return torch.from_numpy(h)

        def store(ptr, t):
            with torch.cuda.device(dev):
                if hip.hipMemcpy(ptr, t.data_ptr(), t.numel() * t.element_size(), H2D) != 0:
                    raise RuntimeError("hipMemcpy H2D")

        def allgather(user, send, recv, nbytes, stream):
            try:
                h = fetch(send, nbytes, stream)
                out = torch.empty(world * nbytes, dtype=torch.uint8)
                dist.all_gather_into_tensor(out, h, group=group)
                store(recv, out)
                return 0
            except Exception:  # noqa: BLE001 -- reported through the C return code
                return 1

        def alltoall(user, send, recv, nbytes, stream):
            try:
                h = fetch(send, world * nbytes, stream)
                out = torch.empty_like(h)
                dist.all_to_all_single(out, h, group=group)
                store(recv, out)
                return 0
            except Exception:  # noqa: BLE001
                return 1

        def allreduce(user, buf, count, op, stream):
            try:
                h = fetch(buf, 4 * count, stream).view(torch.float32)
                dist.all_reduce(h, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM, group=group)
                store(buf, h)
                return 0
            except Exception:  # noqa: BLE001
                return 1

        self._cb = (_lib.ALLGATHER_FN(allgather), _lib.ALLTOALL_FN(alltoall), _lib.ALLREDUCE_F32_FN(allreduce))  # kept alive
