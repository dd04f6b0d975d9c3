import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from influentialrs_amd import synth
from influentialrs_amd._lib import IRS_GEMM_F32, IRS_GEMM_H3, IRS_GEMM_X6
from gpu_util import make_engine
B = 176
cfg = synth.make_config("c2"); L = cfg.max_len
sd = synth.irn_state_dict(cfg, 777)
hists = synth.user_histories(B, cfg.n_item, seed=41)
rows = synth.eval_rows(hists, cfg.n_item, seed=43)
_, seqs, users, _, _ = synth.collate_eval_irs(rows, L, gap_len=1)
seq, u = torch.from_numpy(seqs).cuda(), torch.from_numpy(users).cuda()
pos = torch.full((B,), L - 2, dtype=torch.int32, device="cuda")
eng = make_engine(cfg, sd, max_rows=B, max_seqs=B)
ref = {}
for fill in (None, 0xFF, 0x7F, 0x00):
    for full in (True, False):
        for name, mode in (("x6", IRS_GEMM_X6), ("h3", IRS_GEMM_H3), ("f32", IRS_GEMM_F32)):
            eng.decoder_gemm = mode
            if fill is not None:
                eng._ws.fill_(fill)
            x, xr, _ = eng.decode(seq, u, want_x=full, pos=pos)
            torch.cuda.synchronize()
            key = (name, full)
            got = (x if full else xr).clone()
            if fill is None:
                ref[key] = got
            else:
                a, b = ref[key], got
                same = torch.equal(torch.isnan(a), torch.isnan(b))
                ok = torch.isfinite(a) & torch.isfinite(b)
                d = float((a - b)[ok].abs().max()) if ok.any() else -1
                print("fill %s %s full=%s: nan-pattern equal %s, max diff %g" % (hex(fill), name, full, same, d), flush=True)
