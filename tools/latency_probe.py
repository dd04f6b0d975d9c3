#!/usr/bin/env python3
"""B=1 path-generation latency probe (C2 by default): p50 of a 20-step greedy path, eager vs hipGraph."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from influentialrs_amd import synth
from influentialrs_amd._lib import IRS_MASK_IRN, IRS_SWEEP_BF16, IRS_SWEEP_EXHAUSTIVE, IRS_SWEEP_F32
from influentialrs_amd.engine import Engine
cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
SWEEP = {"bf16": IRS_SWEEP_BF16, "f32": IRS_SWEEP_F32, "exh": IRS_SWEEP_EXHAUSTIVE}[sys.argv[4] if len(sys.argv) > 4 else "bf16"]
cfg = synth.make_config(cfgname)
dev = torch.device("cuda:0")
eng = Engine(n_item=cfg.n_item, n_user=cfg.n_user, d=cfg.emb_dim, max_len=cfg.max_len, n_heads=cfg.n_heads, ffn_dim=cfg.ffn_dim,
             n_layers=cfg.n_layers, u_dim=cfg.u_emb_dim, mask_mode=IRS_MASK_IRN, device=dev, max_rows=max(B, int(os.environ.get("PROBE_MAX", "8"))), max_seqs=max(B, int(os.environ.get("PROBE_MAX", "8"))))
eng.bind_state_dict({k: torch.from_numpy(v).to(dev) for k, v in synth.irn_state_dict(cfg, 1234).items()})
wnd = synth.random_windows(B, cfg.max_len, cfg.n_item, seed=3)
if os.environ.get("PROBE_NVALID"):  # windows with exactly this many items (pre-padded, target last)
    nv = min(int(os.environ["PROBE_NVALID"]), cfg.max_len)
    g = np.random.default_rng(5)
    wnd[:] = 0
    wnd[:, cfg.max_len - nv:] = g.integers(1, cfg.n_item + 1, size=(B, nv))
seqs = torch.from_numpy(wnd).to(dev)
print(f"valid tokens per window: {(wnd != 0).sum(axis=1).mean():.0f}")
users = torch.zeros(B, dtype=torch.int64, device=dev)
for graph in (False, True):
    ts = []
    p = torch.zeros((B, 20), dtype=torch.float32, device=dev)
    st = torch.zeros(B, dtype=torch.int32, device=dev)
    work = seqs.clone()
    hep = torch.full((B,), cfg.max_len - 2, dtype=torch.int32, device=dev)
    for it in range(reps + 5):
        work.copy_(seqs); hep.fill_(cfg.max_len - 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.generate_paths(work, users, hep, 20, k=100, sweep=SWEEP, use_graph=graph, paths=p, status=st)
        torch.cuda.synchronize()
        if it >= 5:
            ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{cfgname} B={B} graph={graph}: p50 {np.median(ts):.3f} ms  min {min(ts):.3f} ms  ({np.median(ts)/20*1e3:.0f} us/step)")
