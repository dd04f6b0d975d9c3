"""Lab: where do the split-float16 attention's rows differ most from the float32 attention's?  Two engines over the same
weights (IRS_ATTN_GEMM read at creation), same batch, every layer-kernel mode."""
import os, sys
sys.path.insert(0, ".")
import numpy as np, torch
import bench
from influentialrs_amd import synth
from influentialrs_amd._lib import IRS_GEMM_F32, IRS_GEMM_H3, IRS_GEMM_X6, IRS_MASK_IRN
from influentialrs_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
cfg = synth.make_config("c4", n_item=20000, emb_dim=128, n_heads=4, n_layers=6, n_user=1000)
sd_np = synth.irn_state_dict(cfg, 1234)
sd = {k: torch.from_numpy(v).to(dev) for k, v in sd_np.items()}
def mk(attn):
    os.environ["IRS_ATTN_GEMM"] = attn
    e = Engine(n_item=cfg.n_item, n_user=cfg.n_user, d=cfg.emb_dim, max_len=cfg.max_len, n_heads=cfg.n_heads, ffn_dim=cfg.ffn_dim,
               n_layers=cfg.n_layers, u_dim=cfg.u_emb_dim, mask_mode=IRS_MASK_IRN, device=dev, max_rows=B, max_seqs=B)
    e.bind_state_dict(sd)
    return e
eh, ef = mk("h3"), mk("f32")
seqs = bench.gpu_windows(B, cfg.max_len, cfg.n_item, dev, seed=3)
users = torch.randint(0, cfg.n_user, (B,), device=dev)
pos = torch.full((B,), cfg.max_len - 2, dtype=torch.int32, device=dev)
for mode, name in ((IRS_GEMM_X6, "x6"), (IRS_GEMM_H3, "h3")):
    eh.decoder_gemm = mode
    ef.decoder_gemm = mode
    a = eh.decode(seqs, users, want_x=False, pos=pos)[1]
    b = ef.decode(seqs, users, want_x=False, pos=pos)[1]
    ef.decoder_gemm = IRS_GEMM_F32
    c = ef.decode(seqs, users, want_x=False, pos=pos)[1]
    for nm, x, y in (("attn h3 vs attn f32 (layer %s)" % name, a, b), ("attn f32, layer %s vs layer f32" % name, b, c)):
        dd = (x - y).abs()
        u = int(dd.max(dim=1).values.argmax())
        print("%s: max %.3g mean %.3g; worst user %d (valid tokens %d), its row max |x| %.3g, its 5 largest diffs %s" % (
            nm, float(dd.max()), float(dd.mean()), u, int((seqs[u] != 0).sum()), float(y[u].abs().max()),
            [round(float(v), 6) for v in dd[u].topk(5).values]))
    rows_bad = (dd.max(dim=1).values > 1e-4).sum()
    print("   rows with a diff > 1e-4:", int(rows_bad))
