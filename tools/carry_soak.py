#!/usr/bin/env python3
"""Soak of the carried emission thresholds inside the library's own search loop: irs_generate_paths, 20 steps, 1024 users on the
C3 catalog (1M x 128), several window seeds -- an engine with the carry (default: pre-pass + selection every 8th step) against one
created with IRS_THR_CARRY=0 (every step): the paths must be identical id for id; time per step of both.
usage: python tools/carry_soak.py [seeds=6]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import bench
from influentialrs_amd import synth
from influentialrs_amd._lib import IRS_SWEEP_BF16
from gpu_util import make_engine

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
cfg = synth.make_config("c3")
B, P = 1024, 20
sd = synth.irn_state_dict(cfg, 1234)
eng_c = make_engine(cfg, sd, max_rows=B, max_seqs=B)
os.environ["IRS_THR_CARRY"] = "0"
eng_n = make_engine(cfg, sd, max_rows=B, max_seqs=B)
del os.environ["IRS_THR_CARRY"]
tc = tn = 0.0
for seed in range(ns):
    seqs = bench.gpu_windows(B, cfg.max_len, cfg.n_item, dev, seed=500 + seed)
    users = torch.randint(0, cfg.n_user, (B,), device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
    hep = torch.full((B,), cfg.max_len - 2, dtype=torch.int32, device=dev)
    out = []
    for eng in (eng_c, eng_n):
        s_, h_ = seqs.clone(), hep.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p, st = eng.generate_paths(s_, users, h_, P, k=100, sweep=IRS_SWEEP_BF16)
        torch.cuda.synchronize()
        out.append((p, st, time.perf_counter() - t0))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]), seed
    if seed:  # (the first search warms both engines up)
        tc += out[0][2]
        tn += out[1][2]
n = max(ns - 1, 1)
print("carry soak: %d searches of %d users x %d steps on 1M x 128: paths identical; %.3f ms per step with the carry, %.3f without" % (
    ns, B, P, tc / n / P * 1e3, tn / n / P * 1e3))
