"""Deterministic synthetic weights and ml-1m-shaped evaluation data.

Nothing here comes from the reference's data files (MovieLens may not be
redistributed); shapes follow SURVEY.md section 8 (D2) and BASELINE.md section 3.

Weights follow torch's default init *families* (Embedding ~ N(0,1) with a zero
pad row, Linear ~ U(+-1/sqrt(fan_in)), MHA in-proj Xavier-uniform, LayerNorm
(1, 0)) except that every bias is drawn N(0, 0.1) so the bias paths and the
constant cross-attention vector (reference model/influentialRS.py:172-173,
SURVEY fact 7) are exercised.  The generator is counter based (numpy Philox,
one stream per tensor name) so any subset of tensors can be regenerated
bit-identically on any box with this numpy.
"""
from __future__ import annotations

import argparse
import hashlib
import math
from typing import Dict, List, Optional, Tuple

import numpy as np

# (n_user, n_item, d, L, H, layers, ffn, u_d)  -- BASELINE.md section 3 table
CONFIGS = {
    # G-tiny of SURVEY 8c: full tensors fit in a fixture
    "tiny": dict(n_user=64, n_item=257, emb_dim=16, max_len=12, n_heads=2, n_layers=2, ffn_dim=32, u_emb_dim=10),
    # reference CLI defaults (main.py:26,36-40); d=30, hd=5
    "default": dict(n_user=6040, n_item=3415, emb_dim=30, max_len=60, n_heads=6, n_layers=6, ffn_dim=256, u_emb_dim=10),
    "c1": dict(n_user=6040, n_item=3415, emb_dim=64, max_len=50, n_heads=4, n_layers=6, ffn_dim=256, u_emb_dim=10),
    "c2": dict(n_user=6040, n_item=3415, emb_dim=128, max_len=200, n_heads=4, n_layers=6, ffn_dim=256, u_emb_dim=10),
    "c3": dict(n_user=100_000, n_item=1_000_000, emb_dim=128, max_len=200, n_heads=4, n_layers=6, ffn_dim=256, u_emb_dim=10),
    "c4": dict(n_user=100_000, n_item=10_000_000, emb_dim=256, max_len=200, n_heads=8, n_layers=6, ffn_dim=256, u_emb_dim=10),
    # C4 / C5's decoder shape (d = 256, 8 heads) on an ml-1m-sized catalog: small enough for the reference to run
    # on the CPU in seconds per user, so the d = 256 kernels get a reference-run golden of their own (irn_c4d)
    "c4d": dict(n_user=6040, n_item=3415, emb_dim=256, max_len=200, n_heads=8, n_layers=6, ffn_dim=256, u_emb_dim=10),
    # evaluator (SampleNet) defaults, model_params.py:116-126
    "eval_default": dict(n_user=6040, n_item=3415, emb_dim=30, max_len=60, n_heads=6, n_layers=6, ffn_dim=120, u_emb_dim=10),
    "eval_tiny": dict(n_user=64, n_item=257, emb_dim=16, max_len=12, n_heads=2, n_layers=2, ffn_dim=24, u_emb_dim=10),
}


def make_config(name: str, **overrides) -> argparse.Namespace:
    """Namespace with the attribute names InfluentialNet.__init__ reads
    (reference model/influentialRS.py:36-47,90)."""
    cfg = dict(CONFIGS[name])
    cfg.update(dropout=0.05, lr1=3e-3, name=name)
    cfg.update(overrides)
    return argparse.Namespace(**cfg)


def _stream(seed: int, name: str) -> np.random.Generator:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    key = int.from_bytes(h[:16], "little")
    return np.random.Generator(np.random.Philox(key=key))


def _normal(seed, name, shape, std=1.0):
    return (_stream(seed, name).standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)


def _uniform(seed, name, shape, bound):
    g = _stream(seed, name)
    return ((g.random(shape, dtype=np.float32) * 2.0 - 1.0) * np.float32(bound)).astype(np.float32)


def positional_encoding(d_model: int, max_len: int) -> np.ndarray:
    """Sinusoidal table, [1, max_len, d] float32 (reference model/layers.py:17-32).
    Computed in float32 like the reference (exp/sin/cos of float32 arrays)."""
    pe = np.zeros((max_len, d_model), dtype=np.float32)
    position = np.arange(0, max_len, dtype=np.float32)[:, None]
    div_term = np.exp(np.arange(0, d_model, 2).astype(np.float32) * np.float32(-math.log(10000.0) / d_model)).astype(np.float32)
    pe[:, 0::2] = np.sin(position * div_term)
    pe[:, 1::2] = np.cos(position * div_term)
    return pe[None]


def irn_state_dict(cfg, seed: int = 1234, evaluator: bool = False) -> Dict[str, np.ndarray]:
    """state_dict (numpy float32) with exactly the reference's key set
    (SURVEY section 5 'Checkpoint / resume').  evaluator=True gives the SampleNet
    key set (word_embedder instead of item_embedder, no user tensors;
    reference model/uRS.py:40-45)."""
    d, F, N, L = cfg.emb_dim, cfg.ffn_dim, cfg.n_item, cfg.max_len
    sd: Dict[str, np.ndarray] = {}
    emb_name = "word_embedder.weight" if evaluator else "item_embedder.weight"
    E = _normal(seed, emb_name, (N + 1, d))
    E[0] = 0.0
    sd[emb_name] = E
    if not evaluator:
        sd["user_embedder.weight"] = _normal(seed, "user_embedder.weight", (cfg.n_user, cfg.u_emb_dim))
        sd["user_mask_layer.weight"] = _uniform(seed, "user_mask_layer.weight", (1, cfg.u_emb_dim), 1.0 / math.sqrt(cfg.u_emb_dim))
        sd["user_mask_layer.bias"] = _normal(seed, "user_mask_layer.bias", (1,), 0.1)
    sd["pos_embedder.pe"] = positional_encoding(d, L)
    sd["project.weight"] = _uniform(seed, "project.weight", (N, d), 1.0 / math.sqrt(d))
    sd["project.bias"] = _normal(seed, "project.bias", (N,), 0.1)
    xav = math.sqrt(6.0 / (d + 3 * d))
    for l in range(cfg.n_layers):
        p = f"decoder.layers.{l}."
        for att in ("self_attn", "multihead_attn"):
            sd[p + att + ".in_proj_weight"] = _uniform(seed, p + att + ".in_proj_weight", (3 * d, d), xav)
            sd[p + att + ".in_proj_bias"] = _normal(seed, p + att + ".in_proj_bias", (3 * d,), 0.1)
            sd[p + att + ".out_proj.weight"] = _uniform(seed, p + att + ".out_proj.weight", (d, d), 1.0 / math.sqrt(d))
            sd[p + att + ".out_proj.bias"] = _normal(seed, p + att + ".out_proj.bias", (d,), 0.1)
        sd[p + "linear1.weight"] = _uniform(seed, p + "linear1.weight", (F, d), 1.0 / math.sqrt(d))
        sd[p + "linear1.bias"] = _normal(seed, p + "linear1.bias", (F,), 0.1)
        sd[p + "linear2.weight"] = _uniform(seed, p + "linear2.weight", (d, F), 1.0 / math.sqrt(F))
        sd[p + "linear2.bias"] = _normal(seed, p + "linear2.bias", (d,), 0.1)
        for n in ("norm1", "norm2", "norm3"):
            sd[p + n + ".weight"] = (1.0 + _normal(seed, p + n + ".weight", (d,), 0.05)).astype(np.float32)
            sd[p + n + ".bias"] = _normal(seed, p + n + ".bias", (d,), 0.05)
    return sd


def user_histories(n_users: int, n_item: int, seed: int = 7, max_hist: int = 2276) -> List[np.ndarray]:
    """ml-1m-shaped histories: lengths log-normal clipped to [18, max_hist],
    median ~95; items Zipf(s~1) without repeats per user;
    ids are 1-based (0 = pad)."""
    g = _stream(seed, "histories")
    lens = np.exp(g.normal(math.log(95.0), 0.95, size=n_users))
    lens = np.clip(lens, 18, min(max_hist, n_item - 2)).astype(np.int64)
    ranks = np.arange(1, n_item + 1, dtype=np.float64)
    p = 1.0 / ranks
    p /= p.sum()
    cdf = np.cumsum(p)
    out = []
    for u in range(n_users):
        need = int(lens[u])
        seen: Dict[int, None] = {}
        while len(seen) < need:
            draw = np.searchsorted(cdf, g.random(2 * (need - len(seen)) + 8)) + 1
            for it in draw:
                it = int(min(it, n_item))
                if it not in seen:
                    seen[it] = None
                    if len(seen) == need:
                        break
        out.append(np.fromiter(seen.keys(), dtype=np.int64, count=need))
    return out


def eval_rows(histories: List[np.ndarray], n_item: int, seed: int = 11, seq_len: int = 100):
    """One evaluation row per user, as DataProvider.get_random_evaluate_data
    builds them (reference data_provider.py:398-449): history = all but the
    last event (last <=100 kept as `raw`), label = last event, target = a
    uniformly random item absent from the kept history."""
    g = _stream(seed, "eval_rows")
    rows = []
    for u, h in enumerate(histories):
        seq, label = h[:-1], int(h[-1])
        raw = seq[-seq_len:]
        hist = set(int(v) for v in raw)
        while True:
            t = int(g.integers(1, n_item + 1))
            if t not in hist:
                break
        rows.append((raw.copy(), u, t, label))
    return rows


def collate_eval_irs(rows, seq_len: int, gap_len: int = 0):
    """The IRN evaluation batch layout (reference DataLoaderEvalIRS._collate_fn,
    data_provider.py:591-617): pre-padded window, gap zeros, target last.
    Returns (raw list, seqs[B,L] int64, users[B], targets[B], labels[B])."""
    B = len(rows)
    seqs = np.zeros((B, seq_len), dtype=np.int64)
    users = np.zeros(B, dtype=np.int64)
    targets = np.zeros(B, dtype=np.int64)
    labels = np.zeros(B, dtype=np.int64)
    raws = []
    l_history = seq_len - gap_len - 1
    for i, (raw, u, t, lab) in enumerate(rows):
        items = raw[-l_history:]
        start = seq_len - len(items) - gap_len - 1
        seqs[i, start:start + len(items)] = items
        seqs[i, -1] = t
        users[i], targets[i], labels[i] = u, t, lab
        raws.append(np.asarray(raw, dtype=np.int64))
    return raws, seqs, users, targets, labels


def random_windows(B: int, L: int, n_item: int, seed: int = 3, min_hist: int = 1):
    """Fast synthetic windows for large-catalog runs (no per-user Python set
    work): pre-padded distinct-ish history, target last."""
    g = _stream(seed, "windows")
    seqs = np.zeros((B, L), dtype=np.int64)
    hl = g.integers(min_hist, L, size=B)
    for b in range(B):
        n = int(hl[b])
        seqs[b, L - 1 - n:L - 1] = g.choice(n_item, size=n, replace=False) + 1 if n_item >= n else g.integers(1, n_item + 1, size=n)
        seqs[b, L - 1] = g.integers(1, n_item + 1)
    return seqs
