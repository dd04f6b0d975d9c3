#!/usr/bin/env python3
"""Time irs_score_topk (and its kernel families) on catalog-scale shapes.
usage: python tools/sweep_bench.py [--shapes N,d,M ...] [--sweep bf16|f32]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from influentialrs_amd import synth
from influentialrs_amd._lib import IRS_MASK_IRN, IRS_PROF_NONE, IRS_PROF_REFINE, IRS_PROF_SWEEP, IRS_SWEEP_BF16, IRS_SWEEP_F32
from influentialrs_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", nargs="*", default=["1000000,128,1024", "1000000,128,32", "1000000,128,1", "1250000,256,1024", "1250000,256,32", "1250000,256,1", "3415,128,1024"])
ap.add_argument("--sweep", default="bf16")
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda:0")
sweep = IRS_SWEEP_BF16 if args.sweep == "bf16" else IRS_SWEEP_F32
engines = {}
for sh in args.shapes:
    N, d, M = map(int, sh.split(","))
    key = (N, d)
    if key not in engines:
        engines.clear()
        torch.cuda.empty_cache()
        nh = d // 32
        cfg = synth.make_config("tiny", n_item=N, emb_dim=d, n_heads=nh, n_layers=1, max_len=4, ffn_dim=8, n_user=2)
        eng = Engine(n_item=N, n_user=2, d=d, max_len=4, n_heads=nh, ffn_dim=8, n_layers=1, u_dim=10, mask_mode=IRS_MASK_IRN,
                     device=dev, max_rows=max(1024, max(int(s.split(",")[2]) for s in args.shapes)), max_seqs=1)
        sd = {k: torch.from_numpy(v).to(dev) for k, v in synth.irn_state_dict(synth.make_config("tiny", n_item=8, emb_dim=d, n_heads=nh, n_layers=1, max_len=4, ffn_dim=8, n_user=2), 1).items()}
        g = torch.Generator(device=dev); g.manual_seed(1)
        sd["item_embedder.weight"] = torch.zeros((N + 1, d), device=dev)
        sd["project.weight"] = (torch.rand((N, d), generator=g, device=dev) * 2 - 1) * d ** -0.5
        sd["project.bias"] = torch.randn(N, generator=g, device=dev) * 0.1
        eng.bind_state_dict(sd)
        engines[key] = eng
    eng = engines[key]
    x = torch.randn((M, d), device=dev)
    for _ in range(2):
        eng.score_topk(x, 100, sweep)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        val, ids, st = eng.score_topk(x, 100, sweep)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / args.reps * 1e6
    fam = {}
    for name, f in (("sweep", IRS_PROF_SWEEP), ("refine", IRS_PROF_REFINE)):
        eng.prof_enable(f)
        for _ in range(args.reps):
            eng.score_topk(x, 100, sweep)
        torch.cuda.synchronize()
        n, ms, fl, by = eng.prof_read()
        fam[name] = (ms / args.reps * 1e3, fl / args.reps, by / args.reps, n / args.reps)
    eng.prof_enable(IRS_PROF_NONE)
    sw_us, fl, by, nl = fam["sweep"]
    alg_flops = 2.0 * d * M * N
    alg_bytes = N * d * 2.0
    print(f"N={N:>8d} d={d:3d} M={M:4d}: total {total:8.1f} us | sweep {sw_us:8.1f} us ({nl:.0f} launches) refine {fam['refine'][0]:7.1f} us | "
          f"alg {alg_flops / total / 1e6:7.1f} TF/s  W-stream {alg_bytes / total / 1e3:7.1f} GB/s | fallback rows {int((st & 1).sum())}")
