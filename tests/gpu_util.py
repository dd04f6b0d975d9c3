"""Helpers shared by the -m gpu tests: build an Engine from synthetic weights."""
import numpy as np
import torch

from influentialrs_amd import synth
from influentialrs_amd.engine import Engine
from influentialrs_amd._lib import IRS_MASK_CAUSAL, IRS_MASK_IRN


def make_engine(cfg, sd_np, *, evaluator=False, max_rows=64, max_seqs=0, rank=0, world=1, max_k=100, device="cuda:0"):
    dev = torch.device(device)
    eng = Engine(n_item=cfg.n_item, n_user=(0 if evaluator else cfg.n_user), d=cfg.emb_dim, max_len=cfg.max_len,
                 n_heads=cfg.n_heads, ffn_dim=cfg.ffn_dim, n_layers=cfg.n_layers,
                 u_dim=(0 if evaluator else cfg.u_emb_dim),
                 mask_mode=(IRS_MASK_CAUSAL if evaluator else IRS_MASK_IRN), device=dev, max_rows=max_rows,
                 max_seqs=max_seqs, max_k=max_k, rank=rank, world=world)
    sd = {k: torch.from_numpy(v).to(dev) for k, v in sd_np.items()}
    eng.bind_state_dict(sd)
    return eng


def scoring_only_engine(n_item, d, W, b, *, max_rows=64, rank=0, world=1, max_k=100, device="cuda:0"):
    """Engine with a 1-layer dummy decoder, for tests of the scoring kernels alone."""
    nh = d // 32 if d % 32 == 0 else 1
    cfg = synth.make_config("tiny", n_item=n_item, emb_dim=d, n_heads=nh, n_layers=1, max_len=4, ffn_dim=8, n_user=2)
    sd = synth.irn_state_dict(cfg, seed=1)
    sd["project.weight"] = W
    sd["project.bias"] = b
    return make_engine(cfg, sd, max_rows=max_rows, max_seqs=1, rank=rank, world=world, max_k=max_k, device=device)
