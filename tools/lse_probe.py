"""(lab) log-sum-exp of a few rows over a small shard: ring form vs register form vs torch; usage: python tools/lse_probe.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from influentialrs_amd import synth
from gpu_util import make_engine

for d, H, n_item, M in ((128, 4, 900, 9), (128, 4, 5000, 32), (256, 8, 900, 9), (256, 8, 70000, 32)):
    cfg = synth.make_config("tiny", emb_dim=d, n_heads=H, ffn_dim=32, max_len=40, n_layers=1, n_item=n_item, n_user=9)
    sd = synth.irn_state_dict(cfg, 77)
    W = torch.from_numpy(sd["project.weight"]).cuda()
    b = torch.from_numpy(sd["project.bias"]).cuda()
    x = torch.randn(M, d, device="cuda") * 0.5
    ref = torch.logsumexp((x.double() @ W.double().T + b.double()), dim=1)
    for ring in ("1", "0"):
        os.environ["IRS_LSE_RING"] = ring
        eng = make_engine(cfg, sd, max_rows=M, max_seqs=M)
        val, ids, st, mx, sm = eng.score_topk_lse(x, 10)
        got = (mx.double() + sm.double().log())
        print(f"d={d} n_item={n_item} M={M} ring={ring}: max |lse - ref| = {(got - ref).abs().max().item():.3e}  rows off: "
              f"{((got - ref).abs() > 1e-4).nonzero().flatten().tolist()}", flush=True)
