#!/usr/bin/env python3
"""Lab: how far a row's k-th best score moves from one path-search step to the next (would an emission threshold carried
from step t serve step t + 1?).  C3-shaped (1M items, d = 128) and C2 windows, 1024 users, 8 greedy steps: per step the exact
100th and 400th best scores of every row (torch matmul + topk), then for each step the share of rows whose new 100th score lies
below the previous step's 100th (threshold too high: a miss) and below the previous 400th / 1600th score (the margin a
carried threshold would need)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench
from influentialrs_amd import synth
from influentialrs_amd._lib import IRS_SWEEP_BF16
from gpu_util import make_engine
dev = torch.device("cuda:0")
for cfgname in ("c3", "c2"):
    cfg = synth.make_config(cfgname)
    B = 1024
    sd = synth.irn_state_dict(cfg, 1234)
    eng = make_engine(cfg, sd, max_rows=B, max_seqs=B)
    W = torch.from_numpy(sd["project.weight"]).to(dev)
    bias = torch.from_numpy(sd["project.bias"]).to(dev)
    seqs = bench.gpu_windows(B, cfg.max_len, cfg.n_item, dev, seed=100)
    users = torch.randint(0, cfg.n_user, (B,), device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    hep = torch.full((B,), cfg.max_len - 2, dtype=torch.int32, device=dev)
    paths = torch.zeros((B, 1), dtype=torch.float32, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    prev = None
    for step in range(8):
        _, xr, _ = eng.decode(seqs, users, want_x=False, pos=hep)
        sc = torch.empty((B, cfg.n_item), dtype=torch.float32, device=dev)
        for c0 in range(0, B, 128):
            sc[c0:c0 + 128] = xr[c0:c0 + 128] @ W.t() + bias
        kk = min(1600, cfg.n_item)
        top = sc.topk(kk, dim=1).values
        a100, a400, a1600 = top[:, 99], top[:, min(399, kk - 1)], top[:, kk - 1]
        if prev is not None:
            p100, p400, p1600 = prev
            print("%s step %d: rows whose 100th score fell below the previous 100th: %.1f %%, below the previous 400th: %.2f %%, below the previous %dth: %.2f %%;  "
                  "median |move| of the 100th score %.3g vs median gap 100th - 400th %.3g" % (
                      cfgname, step, 100 * float((a100 < p100).float().mean()), 100 * float((a100 < p400).float().mean()), kk,
                      100 * float((a100 < p1600).float().mean()), float((a100 - p100).abs().median()), float((a100 - a400).median())), flush=True)
        prev = (a100.clone(), a400.clone(), a1600.clone())
        val, ids, st = eng.score_topk(xr, 100, IRS_SWEEP_BF16)
        eng.path_step(seqs, hep, val, ids, 0, paths, status)
    del eng, W, sc
    torch.cuda.empty_cache()
