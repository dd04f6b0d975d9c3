#!/usr/bin/env python3
"""Soak: many random row batches through irs_score_topk / irs_score_topk_lse at catalog scale (fallback flags, a few
rows per batch against the exhaustive kernel), and many C2 path searches (status flags, ids in range).
usage: python tools/stress.py [batches] [searches]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from influentialrs_amd import synth
from influentialrs_amd._lib import IRS_SWEEP_BF16, IRS_SWEEP_EXHAUSTIVE
from gpu_util import make_engine, scoring_only_engine
import bench

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(5)
for (N, d) in ((1_000_000, 128), (1_250_000, 256)):
    W = ((torch.rand((N, d), generator=g, device=dev) * 2 - 1) * d ** -0.5).cpu().numpy()
    b = (torch.randn(N, generator=g, device=dev) * 0.1).cpu().numpy()
    eng = scoring_only_engine(N, d, W, b, max_rows=1024)
    fb = 0
    for it in range(nb):
        M = int(torch.randint(1, 1025, (1,), generator=g, device=dev).item())
        scale = float(torch.rand(1, generator=g, device=dev).item() * 4 + 0.25)
        x = torch.randn((M, d), generator=g, device=dev) * scale
        if it % 2:
            v, i, st, mx, sm = eng.score_topk_lse(x, 100, IRS_SWEEP_BF16)
        else:
            v, i, st = eng.score_topk(x, 100, IRS_SWEEP_BF16)
        rows = torch.randint(0, M, (min(M, 3),), generator=g, device=dev)
        ev, ei, _ = eng.score_topk(x[rows].contiguous(), 100, IRS_SWEEP_EXHAUSTIVE)
        torch.cuda.synchronize()
        assert torch.equal(i[rows], ei) and torch.equal(v[rows].view(torch.int32), ev.view(torch.int32)), (N, d, it)
        fb += int((st & 1).sum().item())
    print(f"N={N} d={d}: {nb} batches ok, fallback rows {fb}", flush=True)
    del eng

# near-duplicate items (clusters of ~1000 items a few 1e-3 apart: the bf16 filter cannot tell them apart, the
# candidate buffers fill, rows fall back to the exhaustive path) -- results must stay exact
N, d = 400_000, 128
centers = torch.randn((400, d), generator=g, device=dev) * d ** -0.5
W = (centers[torch.randint(0, 400, (N,), generator=g, device=dev)] + torch.randn((N, d), generator=g, device=dev) * 2e-3).cpu().numpy()
b = (torch.randn(N, generator=g, device=dev) * 0.01).cpu().numpy()
eng = scoring_only_engine(N, d, W, b, max_rows=256)
fb = 0
for it in range(max(nb // 4, 4)):
    M = int(torch.randint(1, 257, (1,), generator=g, device=dev).item())
    x = torch.randn((M, d), generator=g, device=dev)
    v, i, st = eng.score_topk(x, 100, IRS_SWEEP_BF16)
    rows = torch.randint(0, M, (min(M, 3),), generator=g, device=dev)
    ev, ei, _ = eng.score_topk(x[rows].contiguous(), 100, IRS_SWEEP_EXHAUSTIVE)
    torch.cuda.synchronize()
    assert torch.equal(i[rows], ei) and torch.equal(v[rows].view(torch.int32), ev.view(torch.int32)), ("clustered", it)
    fb += int((st & 1).sum().item())
print(f"clustered catalog N={N}: {max(nb // 4, 4)} batches ok, fallback rows {fb}", flush=True)
del eng

cfg = synth.make_config("c2")
sd = synth.irn_state_dict(cfg, 1234)
B = 512
eng = make_engine(cfg, sd, max_rows=B, max_seqs=B)
bad = 0
for it in range(ns):
    seqs = bench.gpu_windows(B, cfg.max_len, cfg.n_item, dev, seed=1000 + it)
    users = torch.randint(0, cfg.n_user, (B,), device=dev)
    hep = torch.full((B,), cfg.max_len - 2, dtype=torch.int32, device=dev)
    paths, status = eng.generate_paths(seqs, users, hep, 20, use_graph=bool(it & 1))
    torch.cuda.synchronize()
    p = paths.cpu().numpy()
    assert np.isfinite(p).all() and (p >= 1).all() and (p <= cfg.n_item).all(), it
    for r in range(0, B, 37):  # no item twice in a path unless the window forgot it (L = 200 > 20: never)
        assert len(set(p[r].tolist())) == 20, (it, r, p[r])
    bad += int((status != 0).sum().item())
print(f"c2: {ns} x {B} path searches ok, rows with a status flag {bad}", flush=True)
