#!/usr/bin/env python3
"""Decoder rows with the fused layer kernel on float32 MFMAs vs on split-bf16 MFMAs (IRS_DECODER_GEMM=x6), same inputs:
row error against each other and against the float64-free oracle is printed by the tests; this probe prints the difference
between the two paths and the step time.  usage: python tools/x6_probe.py [users]"""
import os, subprocess, sys
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512

if len(sys.argv) > 2:  # child: run one mode, dump rows
    import time
    import torch
    import bench
    from influentialrs_amd import synth
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from gpu_util import make_engine
    cfg = synth.make_config("c2")
    eng = make_engine(cfg, synth.irn_state_dict(cfg, 1234), max_rows=B, max_seqs=B)
    dev = torch.device("cuda:0")
    seqs = bench.gpu_windows(B, cfg.max_len, cfg.n_item, dev, seed=3)
    users = torch.randint(0, cfg.n_user, (B,), device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    pos = torch.full((B,), cfg.max_len - 2, dtype=torch.int32, device=dev)
    x, xr, _ = eng.decode(seqs, users, want_x=True, pos=pos)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.decode(seqs, users, want_x=False, pos=pos)
    torch.cuda.synchronize()
    print(f"{sys.argv[2]}: decode {1e3 * (time.perf_counter() - t0) / 10:.3f} ms, nan rows {int(torch.isnan(xr).any(1).sum())}", flush=True)
    np.save(sys.argv[3], xr.cpu().numpy())
    sys.exit(0)

outs = {}
for mode in ("f32", "x6"):
    f = f"/tmp/x6_probe_{mode}.npy"
    env = dict(os.environ, IRS_DECODER_GEMM=mode)
    subprocess.check_call([sys.executable, __file__, str(B), mode, f], env=env)
    outs[mode] = np.load(f)
d = np.abs(outs["f32"] - outs["x6"])
print(f"rows {B}: max |f32 - x6| = {np.nanmax(d):.3e}, mean {np.nanmean(d):.3e}, nan in x6: {int(np.isnan(outs['x6']).sum())}")
