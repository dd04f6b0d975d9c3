"""Lab: rows-only decode of a throughput batch (default: C4's decoder shape, d = 256, 8 heads, L = 200) in both decoder
arithmetic modes: time per call, difference of the consumed rows, and both against the numpy oracle on a few sequences
(not product code).  usage: tools/d256_probe.py [users=1024] [d=256] [layers=6]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import bench
from influentialrs_amd import synth
from influentialrs_amd._lib import IRS_GEMM_F32, IRS_GEMM_H3, IRS_GEMM_X6, IRS_MASK_IRN
from influentialrs_amd.engine import Engine
from oracle import oracle_np as O

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda:0")
cfg = synth.make_config("c4", n_item=20000, emb_dim=d, n_heads=d // 32, n_layers=nl, n_user=1000)
sd_np = synth.irn_state_dict(cfg, 1234)
eng = Engine(n_item=cfg.n_item, n_user=cfg.n_user, d=cfg.emb_dim, max_len=cfg.max_len, n_heads=cfg.n_heads, ffn_dim=cfg.ffn_dim,
             n_layers=cfg.n_layers, u_dim=cfg.u_emb_dim, mask_mode=IRS_MASK_IRN, device=dev, max_rows=B, max_seqs=B)
eng.bind_state_dict({k: torch.from_numpy(v).to(dev) for k, v in sd_np.items()})
seqs = bench.gpu_windows(B, cfg.max_len, cfg.n_item, dev, seed=3)
users = torch.randint(0, cfg.n_user, (B,), device=dev)
pos = torch.full((B,), cfg.max_len - 2, dtype=torch.int32, device=dev)
out = {}
for mode, name in ((IRS_GEMM_X6, "x6"), (IRS_GEMM_H3, "h3"), (IRS_GEMM_F32, "f32"), (IRS_GEMM_X6, "x6"), (IRS_GEMM_H3, "h3")):
    eng.decoder_gemm = mode
    for _ in range(3):
        _, xr, _ = eng.decode(seqs, users, want_x=False, pos=pos)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        _, xr, _ = eng.decode(seqs, users, want_x=False, pos=pos)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per decode of {B} users (d = {d}, {nl} layers)", flush=True)
    out[name] = xr.clone()
for nm in ("x6", "h3"):
    dd = (out[nm] - out["f32"]).abs()
    print("%s vs f32 over the consumed rows: max %.3g  mean %.3g  99.9%% %.3g" % (nm, float(dd.max()), float(dd.mean()), float(dd.flatten().kthvalue(int(dd.numel() * 0.999)).values)))
hs, hu = seqs.cpu().numpy(), users.cpu().numpy()
for b in (0, 1, 2, 3):
    ref = O.decode(sd_np, cfg, hs[b], int(hu[b]))[0][cfg.max_len - 2]
    print("user %d vs numpy oracle: x6 %.3g  h3 %.3g  f32 %.3g" % (b, np.abs(ref - out["x6"][b].cpu().numpy()).max(), np.abs(ref - out["h3"][b].cpu().numpy()).max(), np.abs(ref - out["f32"][b].cpu().numpy()).max()))
