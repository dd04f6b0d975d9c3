"""Lab: distance of the consumed decoder rows to the float32-MFMA kernels (IRS_GEMM_F32, two-kernel path, small float32 chain
everywhere) for the two-kernel path (float32 attention scores) and the sequence-resident launch (float16 plane-product scores),
several batches of 4096 users: mean, p99, p99.9, max of the per-row maximum."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench
from influentialrs_amd import synth
from influentialrs_amd._lib import IRS_GEMM_F32, IRS_GEMM_H3
from gpu_util import make_engine
B = 4096
dev = torch.device("cuda:0")
for cfgname, wseed in (("c2", 1234), ("c2", 77), ("c3", 1234)):
    cfg = synth.make_config(cfgname)
    eng = make_engine(cfg, synth.irn_state_dict(cfg, wseed), max_rows=B, max_seqs=B)
    for seed in (3, 11, 29):
        seqs = bench.gpu_windows(B, cfg.max_len, cfg.n_item, dev, seed=seed)
        users = torch.randint(0, cfg.n_user, (B,), device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
        pos = torch.full((B,), cfg.max_len - 2, dtype=torch.int32, device=dev)
        eng.decoder_seq = 0
        eng.decoder_gemm = IRS_GEMM_F32
        ref = eng.decode(seqs, users, want_x=False, pos=pos)[1].clone()
        eng.decoder_gemm = IRS_GEMM_H3
        two = eng.decode(seqs, users, want_x=False, pos=pos)[1].clone()
        eng.decoder_seq = 1
        seq = eng.decode(seqs, users, want_x=False, pos=pos)[1].clone()
        for nm, x in (("two-kernel", two), ("sequence-resident", seq)):
            rm = (x - ref).abs().max(1).values
            print("%s weights %d windows %2d  %-18s mean %.3g  p99 %.3g  p99.9 %.3g  max %.3g  rows > 4e-5: %d  > 1e-4: %d" % (
                cfgname, wseed, seed, nm, float((x - ref).abs().mean()), float(rm.quantile(0.99)), float(rm.quantile(0.999)), float(rm.max()),
                int((rm > 4e-5).sum()), int((rm > 1e-4).sum())), flush=True)
    del eng
    torch.cuda.empty_cache()
