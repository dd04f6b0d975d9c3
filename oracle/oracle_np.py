"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; influentialrs_amd/ never does.

CPU (numpy, float32) restatement of the reference's hot path, function by
function.  Citations are into /root/reference (read-only, never shipped):

  embed_pe            model/influentialRS.py:174-175, model/layers.py:17-32
  pif                 model/influentialRS.py:180
  irn_mask            model/influentialRS.py:120-155 AS CALLED at :183-184
                      (pi_factor lands in w_h: allowed = r_u, last col = 1.0;
                      SURVEY fact 5) + key padding :171
  causal_mask         model/uRS.py:47-50 + key padding :53
  decoder_layer       torch.nn.TransformerDecoderLayer (post-norm, relu,
                      eps 1e-5) as constructed at influentialRS.py:67-74;
                      memory is all zeros (:172-173) so cross-attention is the
                      constant c_l = W_o b_v + b_o
  decode / forward    model/influentialRS.py:157-216, model/uRS.py:52-69
  accuracy_metrics    model/influentialRS.py:340-390
  get_seq             model/influentialRS.py:392-470 (greedy and top-sample_k)
  eval_*              model/evaluator.py:162-323

Third-party arithmetic: everything above is torch (requirements.txt:9 pins
pytorch=1.12.0; the build container has 2.10.0).  The restatement is pinned by
golden vectors captured from the unmodified reference in the build container
(tests/golden/make_golden.py -> tests/golden/*.npz, checked by
tests/test_oracle_golden.py).  The scoring contraction and every selection on
it use the fixed-order definitions of oracle/oracle_score.c (bit-reproducible).
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    """ctypes handle to oracle/_build/liboracle.so (built by oracle/Makefile)."""
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = ctypes.CDLL(path)
        f32p = ctypes.POINTER(ctypes.c_float)
        i64p = ctypes.POINTER(ctypes.c_int64)
        L.orc_score_chain.argtypes = [f32p, f32p, f32p, ctypes.c_int64, ctypes.c_int, f32p]
        L.orc_score_gather.argtypes = [f32p, f32p, f32p, ctypes.c_int, i64p, ctypes.c_int, f32p]
        L.orc_topk.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, f32p, i64p]
        L.orc_topk.restype = ctypes.c_int
        L.orc_rank.argtypes = [f32p, ctypes.c_int64, ctypes.c_int64, i64p, ctypes.c_int]
        L.orc_rank.restype = ctypes.c_int64
        L.orc_max_sumexp.argtypes = [f32p, ctypes.c_int64, f32p, ctypes.POINTER(ctypes.c_double)]
        L.orc_bf16_rne.argtypes = [ctypes.c_float]
        L.orc_bf16_rne.restype = ctypes.c_uint16
        L.orc_set_threads.argtypes = [ctypes.c_int]
        L.orc_max_threads.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _f32p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i64p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


# --------------------------------------------------------------------------
# scoring contraction + selections (fixed-order definitions, oracle_score.c)
# --------------------------------------------------------------------------
def score_chain(x_row: np.ndarray, W: np.ndarray, b: Optional[np.ndarray]) -> np.ndarray:
    """e[j] = fmaf chain over k ascending seeded with b[j] (project Linear,
    influentialRS.py:214).  x_row [d], W [N,d], b [N] float32 -> [N] float32."""
    x_row = np.ascontiguousarray(x_row, dtype=np.float32)
    W = np.ascontiguousarray(W, dtype=np.float32)
    N, d = W.shape
    out = np.empty(N, dtype=np.float32)
    bp = _f32p(np.ascontiguousarray(b, dtype=np.float32)) if b is not None else None
    lib().orc_score_chain(_f32p(x_row), _f32p(W), bp, N, d, _f32p(out))
    return out


def score_gather(x_row, W, b, ids0: Sequence[int]) -> np.ndarray:
    """Chain scores at selected 0-based items."""
    x_row = np.ascontiguousarray(x_row, dtype=np.float32)
    W = np.ascontiguousarray(W, dtype=np.float32)
    ids = np.ascontiguousarray(ids0, dtype=np.int64)
    out = np.empty(len(ids), dtype=np.float32)
    bp = _f32p(np.ascontiguousarray(b, dtype=np.float32)) if b is not None else None
    lib().orc_score_gather(_f32p(x_row), _f32p(W), bp, W.shape[1], _i64p(ids), len(ids), _f32p(out))
    return out


def topk(scores: np.ndarray, k: int, id_base: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Top-k by (score desc, id asc) (torch .topk(100), influentialRS.py:421).
    Returns (values float32[k'], 0-based ids int64[k']), k' = min(k, N)."""
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    kk = min(k, scores.shape[0])
    val = np.empty(kk, dtype=np.float32)
    idx = np.empty(kk, dtype=np.int64)
    n = lib().orc_topk(_f32p(scores), scores.shape[0], k, id_base, _f32p(val), _i64p(idx))
    return val[:n], idx[:n]


def rank_of(scores: np.ndarray, label0: int, hist0: Sequence[int]) -> int:
    """1-based rank of 0-based item label0 among items not in hist0
    (sort + filter + nonzero, influentialRS.py:375-389)."""
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    h = np.ascontiguousarray(hist0, dtype=np.int64)
    return int(lib().orc_rank(_f32p(scores), scores.shape[0], int(label0), _i64p(h), len(h)))


def max_sumexp(scores: np.ndarray) -> Tuple[float, float]:
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    m = ctypes.c_float()
    s = ctypes.c_double()
    lib().orc_max_sumexp(_f32p(scores), scores.shape[0], ctypes.byref(m), ctypes.byref(s))
    return float(m.value), float(s.value)


def bf16_round(a: np.ndarray) -> np.ndarray:
    """float32 -> bfloat16 (round to nearest even) -> float32."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return (u.astype(np.uint32) << 16).view(np.float32)


# --------------------------------------------------------------------------
# decoder (float32 throughout, like the reference; dtype=np.float64 gives a
# tighter yard-stick when judging which side is off)
# --------------------------------------------------------------------------
def embed_pe(E: np.ndarray, pe: np.ndarray, seq: np.ndarray, dtype=np.float32) -> np.ndarray:
    d = E.shape[1]
    L = seq.shape[0]
    return (E[seq].astype(dtype) * dtype(math.sqrt(d)) + pe.reshape(-1, d)[:L].astype(dtype)).astype(dtype)


def pif(sd: Dict[str, np.ndarray], user: int, dtype=np.float32) -> float:
    u = sd["user_embedder.weight"][int(user)].astype(dtype)
    w = sd["user_mask_layer.weight"][0].astype(dtype)
    return dtype(np.dot(u, w) + sd["user_mask_layer.bias"].astype(dtype)[0])


def irn_mask(L: int, r_u: float, seq: np.ndarray, dtype=np.float32) -> np.ndarray:
    """Additive [L,L] mask: allowed (j<=i) = r_u, future = -inf, then the
    whole last column = 1.0; then -inf on padded keys."""
    M = np.full((L, L), -np.inf, dtype=dtype)
    i, j = np.tril_indices(L)
    M[i, j] = r_u
    M[:, L - 1] = 1.0
    M[:, seq == 0] = -np.inf
    return M


def causal_mask(L: int, seq: np.ndarray, dtype=np.float32) -> np.ndarray:
    M = np.full((L, L), -np.inf, dtype=dtype)
    i, j = np.tril_indices(L)
    M[i, j] = 0.0
    M[:, seq == 0] = -np.inf
    return M


def layer_norm(z, g, b, eps=1e-5):
    mu = z.mean(axis=-1, keepdims=True)
    var = ((z - mu) ** 2).mean(axis=-1, keepdims=True)
    return (z - mu) / np.sqrt(var + z.dtype.type(eps)) * g + b


def cross_attn_const(sd, prefix: str, d: int, dtype=np.float32) -> np.ndarray:
    """c_l = W_o^{ca} . b_in^{ca}[2d:3d] + b_o^{ca}: attention over an all-zero
    memory returns the value bias for every query (softmax weights sum to 1)."""
    bv = sd[prefix + "multihead_attn.in_proj_bias"][2 * d:3 * d].astype(dtype)
    Wo = sd[prefix + "multihead_attn.out_proj.weight"].astype(dtype)
    bo = sd[prefix + "multihead_attn.out_proj.bias"].astype(dtype)
    return Wo @ bv + bo


def decoder_layer(sd, prefix: str, x: np.ndarray, mask: np.ndarray, H: int) -> np.ndarray:
    dt = x.dtype.type
    L, d = x.shape
    hd = d // H
    Win = sd[prefix + "self_attn.in_proj_weight"].astype(dt)
    bin_ = sd[prefix + "self_attn.in_proj_bias"].astype(dt)
    qkv = x @ Win.T + bin_
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    o = np.empty_like(x)
    scale = dt(1.0 / math.sqrt(hd))
    for h in range(H):
        sl = slice(h * hd, (h + 1) * hd)
        s = (q[:, sl] * scale) @ k[:, sl].T + mask
        m = s.max(axis=1, keepdims=True)
        p = np.exp(s - m)
        p = p / p.sum(axis=1, keepdims=True)
        o[:, sl] = p @ v[:, sl]
    Wo = sd[prefix + "self_attn.out_proj.weight"].astype(dt)
    bo = sd[prefix + "self_attn.out_proj.bias"].astype(dt)
    x = layer_norm(x + (o @ Wo.T + bo), sd[prefix + "norm1.weight"].astype(dt), sd[prefix + "norm1.bias"].astype(dt))
    c = cross_attn_const(sd, prefix, d, dt)
    x = layer_norm(x + c, sd[prefix + "norm2.weight"].astype(dt), sd[prefix + "norm2.bias"].astype(dt))
    W1 = sd[prefix + "linear1.weight"].astype(dt)
    b1 = sd[prefix + "linear1.bias"].astype(dt)
    W2 = sd[prefix + "linear2.weight"].astype(dt)
    b2 = sd[prefix + "linear2.bias"].astype(dt)
    f = np.maximum(x @ W1.T + b1, 0) @ W2.T + b2
    x = layer_norm(x + f, sd[prefix + "norm3.weight"].astype(dt), sd[prefix + "norm3.bias"].astype(dt))
    return x.astype(dt)


def decode(sd, cfg, seq: np.ndarray, user: Optional[int], evaluator: bool = False, dtype=np.float32):
    """One sequence through embed + n_layers decoder layers.
    Returns (x[L,d], r_u or None).  IRN: influentialRS.py:157-200 with the
    as-called mask; evaluator=True: SampleNet.decoding (uRS.py:52-64)."""
    seq = np.asarray(seq, dtype=np.int64)
    L = seq.shape[0]
    E = sd["word_embedder.weight" if evaluator else "item_embedder.weight"]
    x = embed_pe(E, sd["pos_embedder.pe"], seq, dtype)
    if evaluator:
        r_u = None
        mask = causal_mask(L, seq, dtype)
    else:
        r_u = pif(sd, user, dtype)
        mask = irn_mask(L, r_u, seq, dtype)
    with np.errstate(invalid="ignore"):
        for l in range(cfg.n_layers):
            x = decoder_layer(sd, f"decoder.layers.{l}.", x, mask, cfg.n_heads)
    return x, r_u


def forward_logits(sd, cfg, seq, user, evaluator=False, rows: Optional[Sequence[int]] = None, exact=True):
    """logits for the given rows (all L if None): [len(rows), N] float32.
    exact=True uses the fixed-order chain; False a BLAS matmul."""
    x, _ = decode(sd, cfg, seq, user, evaluator)
    W, b = sd["project.weight"], sd["project.bias"]
    if rows is None:
        rows = range(x.shape[0])
    if exact:
        return np.stack([score_chain(x[r], W, b) for r in rows])
    return (x[list(rows)] @ W.T + b).astype(np.float32)


# --------------------------------------------------------------------------
# IRSNN task handlers (one row at a time == the only batch size at which the
# published IRN runs)
# --------------------------------------------------------------------------
def accuracy_metrics(sd, cfg, raw: Sequence[np.ndarray], seqs: np.ndarray, users, labels,
                     top_k=20, gap_len=0, use_h=True, x_rows: Optional[np.ndarray] = None):
    """(hit_count, rr[]) as IRSNN.get_accuracy_metrics_in_batch
    (influentialRS.py:340-390).  rr skips rows whose label is in the filtered
    history (`if label in indices`, :386)."""
    B, L = seqs.shape
    hep = L - (gap_len + 1) - 1
    W, b = sd["project.weight"], sd["project.bias"]
    hit, rr, ranks = 0, [], []
    for i in range(B):
        xr = x_rows[i] if x_rows is not None else decode(sd, cfg, seqs[i], users[i])[0][hep]
        s = score_chain(xr, W, b)
        lab = int(labels[i])
        hist = np.asarray(raw[i], dtype=np.int64) if use_h else np.zeros(0, dtype=np.int64)
        if use_h and lab in set(int(v) for v in hist):
            ranks.append(0)
            continue
        r = rank_of(s, lab - 1, hist - 1)
        ranks.append(r)
        if r <= top_k:
            hit += 1
        rr.append(1.0 / float(r))
    return hit, np.array(rr), np.array(ranks)


def select_next(top_ids1: np.ndarray, top_vals: np.ndarray, window: np.ndarray, sample=False,
                sample_k=3, rng: Optional[np.random.Generator] = None) -> int:
    """First of the top-100 (1-based ids, descending) not in the window
    (influentialRS.py:423-434).  Raises IndexError like the reference when all
    100 are in the window."""
    present = np.isin(top_ids1, window)
    surv = top_ids1[~present]
    if not sample:
        return int(surv[0])
    sv = top_vals[~present][:sample_k].astype(np.float64)
    p = np.exp(sv - sv.max())
    p /= p.sum()
    return int(surv[:sample_k][rng.choice(len(p), p=p)])


def get_seq(sd, cfg, seqs: np.ndarray, users, targets, max_path_len=20, gap_len=0, k_cand=100,
            return_trace=False):
    """Greedy persuasion-path generation, IRSNN.get_seq_in_batch
    (influentialRS.py:392-470) with per-row history_end_pos (the reference
    shares one int across rows, which only matters for gap_len>0 and B>1;
    SURVEY Appendix B).  Returns (paths float32 [B,P], targets, histories, n_early[, trace])."""
    seqs = np.asarray(seqs, dtype=np.int64)
    B, L = seqs.shape
    W, b = sd["project.weight"], sd["project.bias"]
    paths = np.zeros((B, max_path_len), dtype=np.float32)
    trace = []
    for r in range(B):
        win = seqs[r].copy()
        hep = L - (gap_len + 1) - 1
        for i in range(max_path_len):
            x, _ = decode(sd, cfg, win, users[r])
            s = score_chain(x[hep], W, b)
            vals, ids0 = topk(s, k_cand)
            nxt = select_next(ids0 + 1, vals, win[:hep + 1])
            if return_trace:
                trace.append((r, i, x[hep].copy(), vals.copy(), ids0.copy(), nxt))
            paths[r, i] = nxt
            if hep < L - 2:
                win[hep + 1] = nxt
                hep += 1
            else:
                new = np.zeros(L, dtype=np.int64)
                new[:-2] = win[1:-1]
                new[-2] = nxt
                new[-1] = win[-1]
                win = new
    n_early = 0
    histories = []
    tg = np.asarray(targets, dtype=np.int64)
    for r in range(B):
        pos = np.where(paths[r] == tg[r])[0]
        if len(pos):
            n_early += 1
            paths[r, pos[0] + 1:] = 0
        h = seqs[r, :-1]
        histories.append(h[h != 0])
    if return_trace:
        return paths, tg, histories, n_early, trace
    return paths, tg, histories, n_early


def beam_search(sd, cfg, seqs: np.ndarray, users, max_path_len=20, gap_len=0, beam=4, k_cand=100):
    """BUILD-DEFINED beam search (the reference has none: SURVEY fact 4, row A9) -- the CPU
    statement of influentialrs_amd/csrc/path.hip:k_beam_step, PARITY UNPINNED by the reference
    for beam > 1; beam == 1 is get_seq() id for id.
    Per step, every live beam contributes its first `beam` window-survivors of its top-k_cand,
    scored cum + log-softmax(item); the best `beam` by (score desc, parent asc, rank asc) survive.
    Returns (paths float32 [B, beam, P], scores float64 [B, beam])."""
    seqs = np.asarray(seqs, dtype=np.int64)
    B, L = seqs.shape
    W, b = sd["project.weight"], sd["project.bias"]
    paths = np.zeros((B, beam, max_path_len), dtype=np.float32)
    scores = np.full((B, beam), -np.inf)
    for r in range(B):
        beams = [(0.0, seqs[r].copy(), L - (gap_len + 1) - 1, [])]
        for step in range(max_path_len):
            cands = []
            for j, (cum, win, hep, path) in enumerate(beams):
                x, _ = decode(sd, cfg, win, users[r])
                s = score_chain(x[hep], W, b)
                vals, ids0 = topk(s, k_cand)
                norm = 0.0
                if beam > 1:
                    m, se = max_sumexp(s)
                    norm = m + math.log(se)
                present = np.isin(ids0 + 1, win[:hep + 1])
                sv, si = vals[~present][:beam], (ids0 + 1)[~present][:beam]
                for rank, (v, it) in enumerate(zip(sv, si)):
                    cands.append((cum + (float(v) - norm), j, rank, int(it)))
            cands.sort(key=lambda c: (-c[0], c[1], c[2]))
            new = []
            for sc, j, rank, it in cands[:beam]:
                cum, win, hep, path = beams[j]
                if hep < L - 2:
                    nw = win.copy()
                    nw[hep + 1] = it
                    nh = hep + 1
                else:
                    nw = np.zeros(L, dtype=np.int64)
                    nw[:-2] = win[1:-1]
                    nw[-2] = it
                    nw[-1] = win[-1]
                    nh = hep
                new.append((sc, nw, nh, path + [it]))
            beams = new
        for j, (sc, _, _, path) in enumerate(beams):
            paths[r, j, :len(path)] = path
            scores[r, j] = sc
    return paths, scores


# --------------------------------------------------------------------------
# Evaluator side (SampleNet + Evaluator, model/uRS.py, model/evaluator.py)
# --------------------------------------------------------------------------
def _first_none_zero_index(t: np.ndarray) -> int:
    """evaluator.py:136-144 (name as in the reference: index before the first 0)."""
    z = np.where(t == 0)[0]
    return len(t) - 1 if len(z) == 0 else int(z[0]) - 1


def _last_path_index(t: np.ndarray, target: int) -> int:
    """evaluator.py:146-154."""
    p = np.where(t == target)[0]
    return _first_none_zero_index(t) if len(p) == 0 else int(p[0]) - 1


def _log_softmax_at(s: np.ndarray, ids0: Sequence[int]) -> np.ndarray:
    m, se = max_sumexp(s)
    return np.array([float(s[j]) - m - math.log(se) for j in ids0], dtype=np.float64)


def eval_get_pp(sd, cfg, new_seqs: np.ndarray, start_pos, l_paths) -> List[float]:
    """Evaluator.get_pp_in_batch (evaluator.py:292-323): mean NLL over the path
    positions of each sequence."""
    W, b = sd["project.weight"], sd["project.bias"]
    out = []
    for i in range(new_seqs.shape[0]):
        inp = new_seqs[i, :-1]
        x, _ = decode(sd, cfg, inp, None, evaluator=True)
        left = int(start_pos[i])
        right = left + int(l_paths[i])
        tgt = new_seqs[i, left:right]
        nll = []
        for pos, t in zip(range(left - 1, right - 1), tgt):
            if t <= 0:
                continue
            s = score_chain(x[pos], W, b)
            nll.append(-_log_softmax_at(s, [int(t) - 1])[0])
        out.append(float(np.mean(nll)))
    return out


def eval_get_rr_increase(sd, cfg, histories: np.ndarray, new_seqs: np.ndarray, targets):
    """Evaluator.get_rr_increase_in_batch (evaluator.py:245-290)."""
    W, b = sd["project.weight"], sd["project.bias"]
    begin_r, end_r = [], []
    for i in range(histories.shape[0]):
        t = int(targets[i])
        inp = histories[i, :-1]
        end = _first_none_zero_index(inp)
        x, _ = decode(sd, cfg, inp, None, evaluator=True)
        s = score_chain(x[end], W, b)
        begin_r.append(rank_of(s, t - 1, inp[:end + 1] - 1))
        inp = new_seqs[i, :-1]
        end = _last_path_index(inp, t)
        x, _ = decode(sd, cfg, inp, None, evaluator=True)
        s = score_chain(x[end], W, b)
        end_r.append(rank_of(s, t - 1, inp[:end + 1] - 1))
    irr = np.array([1 / end_r[i] - 1 / begin_r[i] for i in range(len(end_r))])
    ir = np.array([end_r[i] - begin_r[i] for i in range(len(end_r))])
    return irr, ir


def eval_get_grad(sd, cfg, histories: np.ndarray, new_seqs: np.ndarray, targets, start_pos, l_paths):
    """Evaluator.get_grad_in_batch (evaluator.py:162-243) on a private copy of
    `histories` (the reference mutates its argument through views)."""
    W, b = sd["project.weight"], sd["project.bias"]
    B = histories.shape[0]
    paths = [new_seqs[i, int(start_pos[i]):int(start_pos[i]) + int(l_paths[i])] for i in range(B)]
    S = int(max(l_paths))
    wins = histories[:, :-1].copy()
    Lw = wins.shape[1]
    t_probs = np.zeros((B, S))
    p_probs = np.zeros((B, S))
    for i in range(S):
        for j in range(B):
            end = _first_none_zero_index(wins[j])
            if i < int(l_paths[j]):
                nxt = int(paths[j][i])
                x, _ = decode(sd, cfg, wins[j], None, evaluator=True)
                s = score_chain(x[end], W, b)
                lp = _log_softmax_at(s, [nxt - 1, int(targets[j]) - 1])
                p_probs[j, i], t_probs[j, i] = lp[0], lp[1]
            else:
                nxt = 0
            if end == Lw - 1:
                new = np.zeros(Lw, dtype=np.int64)
                new[:-1] = wins[j][1:]
                new[-1] = nxt
                wins[j] = new
            else:
                wins[j][end + 1] = nxt
    avg_ps, iois = [], []
    for i in range(B):
        tp = t_probs[i][t_probs[i] < 0]
        pp = p_probs[i][p_probs[i] < 0]
        iois.append(tp[-1] - tp[0])
        avg_ps.append(sum(pp) / len(pp))
    return t_probs, p_probs, avg_ps, iois
