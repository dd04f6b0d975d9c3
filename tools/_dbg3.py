import sys, os, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from influentialrs_amd import synth
from influentialrs_amd._lib import IRS_GEMM_F32, IRS_GEMM_H3, IRS_GEMM_X6, IRS_SWEEP_F32
from gpu_util import make_engine
B = 176
cfg = synth.make_config("c2"); L = cfg.max_len
sd = synth.irn_state_dict(cfg, 777)
hists = synth.user_histories(B, cfg.n_item, seed=41)
rows = synth.eval_rows(hists, cfg.n_item, seed=43)
_, seqs, users, _, _ = synth.collate_eval_irs(rows, L, gap_len=1)
seq, u = torch.from_numpy(seqs).cuda(), torch.from_numpy(users).cuda()
pos = torch.full((B,), L - 2, dtype=torch.int32, device="cuda")
ref = {}
bad = 0
t0 = time.time()
it = 0
while time.time() - t0 < 100:
    eng = make_engine(cfg, sd, max_rows=B, max_seqs=B)
    for name, mode in (("x6", IRS_GEMM_X6), ("h3", IRS_GEMM_H3), ("f32", IRS_GEMM_F32)):
        eng.decoder_gemm = mode
        x, xr, _ = eng.decode(seq, u, want_x=True, pos=pos)
        val, ids, st = eng.score_topk(xr, 100, IRS_SWEEP_F32)
        torch.cuda.synchronize()
        got = x.clone()
        if name not in ref:
            ref[name] = got
        else:
            a = ref[name]
            ok = torch.isfinite(a) & torch.isfinite(got)
            d = float((a - got)[ok].abs().max())
            if d != 0.0 or not torch.equal(torch.isnan(a), torch.isnan(got)):
                bad += 1
                dd = torch.where(ok, (a - got).abs(), torch.zeros_like(a)).max(dim=2).values
                nz = torch.nonzero(dd > 0)
                print("iteration %d mode %s: max diff %g, %d rows differ, first %s" % (it, name, d, nz.shape[0], nz[:6].tolist()), flush=True)
    del eng
    it += 1
print("iterations", it, "non-identical decodes", bad)
