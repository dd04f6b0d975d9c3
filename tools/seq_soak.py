#!/usr/bin/env python3
"""Soak of the sequence-resident decoder launch: many random batches (384 .. 4096 users, ml-1m-shaped windows of fresh seeds,
every tenth batch with uniformly random window lengths 1 .. L) decoded by the sequence-resident launch and by the two-kernel
path; the consumed rows must agree within 1e-4 (their score arithmetic differs), hold no NaN, and the launch must be
deterministic (two runs, same bits).  usage: python tools/seq_soak.py [batches=60]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench
from influentialrs_amd import synth
from gpu_util import make_engine

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
cfg = synth.make_config("c2")
L = cfg.max_len
eng = make_engine(cfg, synth.irn_state_dict(cfg, 1234), max_rows=4096, max_seqs=4096)
g = torch.Generator(device=dev)
g.manual_seed(99)
worst, rows = 0.0, 0
for it in range(nb):
    B = int(torch.randint(384, 4097, (1,), generator=g, device=dev).item())
    seqs = bench.gpu_windows(B, L, cfg.n_item, dev, seed=1000 + it)
    if it % 10 == 9:  # any length 1 .. L (history + target)
        n = torch.randint(1, L + 1, (B, 1), generator=g, device=dev)
        full = seqs.clone()
        full[full == 0] = 7
        seqs = torch.where(torch.arange(L, device=dev)[None, :] >= L - n, full, torch.zeros_like(full))
    users = torch.randint(0, cfg.n_user, (B,), generator=g, device=dev)
    pos = torch.full((B,), L - 2, dtype=torch.int32, device=dev)
    pos[(seqs != 0).sum(1) == 1] = L - 1
    eng.decoder_seq = False
    ref = eng.decode(seqs, users, want_x=False, pos=pos)[1].clone()
    eng.decoder_seq = True
    a = eng.decode(seqs, users, want_x=False, pos=pos)[1].clone()
    b = eng.decode(seqs, users, want_x=False, pos=pos)[1].clone()
    assert eng.decoder_seq_last
    assert not torch.isnan(a).any() and not torch.isnan(ref).any(), it
    assert torch.equal(a, b), ("not deterministic", it)
    d = float((a - ref).abs().max())
    assert d < 1e-4, (it, B, d)
    worst, rows = max(worst, d), rows + B
    if it % 10 == 9:
        print(f"{it + 1} batches, {rows} rows: largest distance between the two paths {worst:.3g}", flush=True)
print(f"sequence-resident soak: {nb} batches, {rows} rows ok; largest distance to the two-kernel path {worst:.3g}")
