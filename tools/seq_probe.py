#!/usr/bin/env python3
"""Lab: the sequence-resident layer kernel (irs_set_decoder_seq) against the default throughput kernels on the same batch:
consumed rows, and the time of a rows-only decode in both modes.  usage: python tools/seq_probe.py [users=4096] [layers=6]"""
import sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np, torch
import os
if os.environ.get("IRS_LAB_LIB"):  # (lab: a SEQ_EXP build of the library, tools/seq_lab.sh)
    from influentialrs_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ["IRS_LAB_LIB"])
import bench
from influentialrs_amd import synth
from gpu_util import make_engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
cfg = synth.make_config("c2", n_layers=nl)
sd = synth.irn_state_dict(cfg, 1234)
eng = make_engine(cfg, sd, max_rows=B, max_seqs=B)
seqs = bench.gpu_windows(B, cfg.max_len, cfg.n_item, dev, seed=3)
if len(sys.argv) > 3:  # every window cut to the same number of tokens (history + target)
    n = int(sys.argv[3])
    full = seqs.clone()
    full[full == 0] = 7
    col = torch.arange(cfg.max_len, device=dev)[None, :]
    seqs = torch.where(col >= cfg.max_len - n, full, torch.zeros_like(full))
users = torch.randint(0, cfg.n_user, (B,), device=dev, generator=torch.Generator(device=dev).manual_seed(1))
pos = torch.full((B,), cfg.max_len - 2, dtype=torch.int32, device=dev)
out = {}
for name, on in (("default", 0), ("seq", 1), ("default", 0), ("seq", 1)):
    eng.decoder_seq = on
    xr = eng.decode(seqs, users, want_x=False, pos=pos)[1].clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.decode(seqs, users, want_x=False, pos=pos)
    torch.cuda.synchronize()
    print("%-8s %.3f ms per decode of %d users (%d layers)" % (name, (time.perf_counter() - t0) * 100, B, nl), flush=True)
    out[name] = xr
if os.environ.get("SEQ_PROBE_F32"):  # both against the float32-MFMA kernels (IRS_GEMM_F32, two-kernel path)
    from influentialrs_amd._lib import IRS_GEMM_F32, IRS_GEMM_H3
    eng.decoder_seq = 0
    eng.decoder_gemm = IRS_GEMM_F32
    ref = eng.decode(seqs, users, want_x=False, pos=pos)[1].clone()
    eng.decoder_gemm = IRS_GEMM_H3
    for nm in ("default", "seq"):
        dd = (out[nm] - ref).abs()
        print("%-8s vs float32-MFMA kernels: max %.3g  mean %.3g  row-max p50 %.3g p99 %.3g p99.9 %.3g" % (nm, float(dd.max()), float(dd.mean()),
              float(dd.max(1).values.quantile(0.5)), float(dd.max(1).values.quantile(0.99)), float(dd.max(1).values.quantile(0.999))))
a, b = out["default"], out["seq"]
nan_a, nan_b = torch.isnan(a).any(1), torch.isnan(b).any(1)
print("NaN rows: default %d, seq %d" % (int(nan_a.sum()), int(nan_b.sum())))
ok = ~(nan_a | nan_b)
diff = (a - b).abs().max(1).values
print("seq vs default over the consumed rows: max %.3g  mean %.3g  rows > 1e-4: %d of %d" % (float(diff[ok].max()), float((a - b)[ok].abs().mean()),
      int((diff[ok] > 1e-4).sum()), int(ok.sum())))
ntok = (seqs != 0).sum(1)
for lo, hi in ((0, 32), (32, 64), (64, 96), (96, 128), (128, 160), (160, 192), (192, 257)):
    sel = (ntok > lo) & (ntok <= hi) & ok
    if int(sel.sum()):
        print("  tokens (%3d, %3d]: %4d rows, bad %4d, max %.3g" % (lo, hi, int(sel.sum()), int((diff[sel] > 1e-4).sum()), float(diff[sel].max())))
bad = torch.nonzero(diff > 1e-4).flatten()[:8].tolist()
for i in bad:
    print("  row %d: tokens %d diff %.3g" % (i, int((seqs[i] != 0).sum()), float(diff[i])))

# ---- which tokens differ: layer 0's attention output (buffer 1) and layer output x' (buffer 0) of both modes, token by token
if nl == 2:
    L = cfg.max_len
    def frag_rows(buf, rows):  # fragment-major [tile][tn 4][g 4][lk 2][li 32][4] -> [rows][128]
        t = buf[: (rows // 32) * 4096].view(rows // 32, 4, 4, 2, 32, 4)       # tile, tn, g, lk, li, e
        return t.permute(0, 4, 1, 2, 3, 5).reshape(rows, 128)               # row = tile * 32 + li; col = 32 tn + 8 g + 4 lk + e
    eng.decoder_seq = 0
    eng.decode(seqs, users, want_x=False, pos=pos)
    torch.cuda.synchronize()
    off = eng.debug_buffer(7, B, torch.int32).clone().long()
    cnt = eng.debug_buffer(8, B, torch.int32).clone().long()
    M = int((off + cnt).max())
    Mp = (M + 31) // 32 * 32
    x_def = frag_rows(eng.debug_buffer(0, Mp * 128, torch.float32), Mp).clone()
    a_def = frag_rows(eng.debug_buffer(1, Mp * 128, torch.float32), Mp).clone()
    eng.decoder_seq = 1
    eng.decode(seqs, users, want_x=False, pos=pos)
    torch.cuda.synchronize()
    nwg = int(eng.debug_buffer(6, 1, torch.int32)[0])
    tseq = eng.debug_buffer(2, nwg * 16, torch.int32).clone().long()   # per half tile: sequence, block index
    tqb = eng.debug_buffer(3, nwg * 16, torch.int32).clone().long()
    R = nwg * 8 * 32
    x_seq = frag_rows(eng.debug_buffer(0, R * 128, torch.float32), R).clone()
    a_seq = frag_rows(eng.debug_buffer(1, R * 128, torch.float32), R).clone()
    print("plan: %d workgroups, %d half tiles used of %d, lane efficiency %.3f" % (nwg, int((tseq >= 0).sum()), nwg * 16, float(cnt.sum()) / R))
    l16 = torch.arange(16, device=dev)
    worst = []
    seen = torch.zeros(int(cnt.sum()), dtype=torch.bool, device=dev)
    for hs in torch.nonzero(tseq >= 0).flatten().tolist():
        b, qb, t, h = int(tseq[hs]), int(tqb[hs]), hs // 2, hs % 2
        j = 16 * qb + l16
        live = j < cnt[b]
        rows_d = (off[b] + j)[live]
        rows_s = (32 * t + 16 * h + l16)[live]
        seen[rows_d] = True
        da = (a_def[rows_d] - a_seq[rows_s]).abs()
        dx = (x_def[rows_d] - x_seq[rows_s]).abs()
        if float(dx.max()) > 1e-4:
            worst.append((t % 8, qb, int(cnt[b]), float(da.max()), float(dx.max()), da.max(1).values, dx.max(1).values))
    print("every token placed exactly once:", bool(seen.all()), int(seen.sum()), "of", int(cnt.sum()))
    print("half tiles with a differing x' token: %d of %d" % (len(worst), int((tseq >= 0).sum())))
    import collections
    print("  by workgroup slot:", dict(collections.Counter(w[0] for w in worst)))
    print("  by tile index in the sequence:", dict(collections.Counter(w[1] for w in worst)))
    for w in worst[:6]:
        print("  slot %d tile %d cnt %d: attention diff %.3g, x' diff %.3g; attention-diff lanes: %s ; heads: " % (w[0], w[1], w[2], w[3], w[4],
              torch.nonzero(w[5] > 1e-5).flatten().tolist()))

