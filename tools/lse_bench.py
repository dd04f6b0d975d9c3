#!/usr/bin/env python3
"""Time irs_score_lse and irs_score_topk_lse (the beam step's scoring) on a float32 catalog.
usage: python tools/lse_bench.py [N] [d] [M] [reps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from influentialrs_amd import synth
from influentialrs_amd._lib import IRS_MASK_IRN, IRS_SWEEP_BF16
from influentialrs_amd.engine import Engine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_500_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
M = int(sys.argv[3]) if len(sys.argv) > 3 else 32
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
nh = d // 32
eng = Engine(n_item=N, n_user=2, d=d, max_len=4, n_heads=nh, ffn_dim=8, n_layers=1, u_dim=10, mask_mode=IRS_MASK_IRN,
             device=dev, max_rows=max(M, 64), max_seqs=1)
small = synth.make_config("tiny", n_item=8, emb_dim=d, n_heads=nh, n_layers=1, max_len=4, ffn_dim=8, n_user=2)
sd = {k: torch.from_numpy(v).to(dev) for k, v in synth.irn_state_dict(small, 1).items()}
g = torch.Generator(device=dev)
g.manual_seed(1)
sd["item_embedder.weight"] = torch.zeros((N + 1, d), device=dev)
sd["project.weight"] = (torch.rand((N, d), generator=g, device=dev) * 2 - 1) * d ** -0.5
sd["project.bias"] = torch.randn(N, generator=g, device=dev) * 0.1
eng.bind_state_dict(sd)
x = torch.randn((M, d), device=dev)
for name, fn in (("score_lse", lambda: eng.score_lse(x)), ("score_topk_lse", lambda: eng.score_topk_lse(x, 100, IRS_SWEEP_BF16)),
                 ("score_topk", lambda: eng.score_topk(x, 100, IRS_SWEEP_BF16))):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    print(f"N={N} d={d} M={M}: {name:15s} {us:9.1f} us   fp32 catalog {N * d * 4 / us / 1e6:7.2f} TB/s-equivalent", flush=True)
