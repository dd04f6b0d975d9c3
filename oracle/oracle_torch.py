"""ORACLE -- TEST / MEASUREMENT INFRASTRUCTURE ONLY.  Not part of the product path.

The same CPU restatement as oracle_np.py (Appendix A of SURVEY.md; reference model/influentialRS.py:157-216 and
:412-450), written with torch CPU operators instead of numpy: the reference's own arithmetic runs on torch's CPU
kernels, so this is the restatement bench.py times as the "reference-equivalent" CPU baseline (B-equiv).  It is
checked against oracle_np.py (which the reference's golden vectors pin) in tests/test_host_logic.py.
Only tests/ and bench.py's cpu_baseline leg import it.
"""
import math

import torch
import torch.nn.functional as F


class TorchIRN:
    """Weights of one IRN as CPU tensors + the as-called forward of a single sequence."""

    def __init__(self, sd_np, cfg):
        self.cfg = cfg
        self.sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
        d = cfg.emb_dim
        self.c = []
        for l in range(cfg.n_layers):
            p = f"decoder.layers.{l}."
            bv = self.sd[p + "multihead_attn.in_proj_bias"][2 * d:3 * d]
            self.c.append(self.sd[p + "multihead_attn.out_proj.weight"] @ bv + self.sd[p + "multihead_attn.out_proj.bias"])

    def decode(self, seq, user):
        """[L] int64 ids, user id -> x[L, d] (as-called mask: allowed = r_u, last column 1.0, pad keys -inf)."""
        sd, cfg = self.sd, self.cfg
        L, d, H = seq.shape[0], cfg.emb_dim, cfg.n_heads
        hd = d // H
        x = sd["item_embedder.weight"][seq] * math.sqrt(d) + sd["pos_embedder.pe"].reshape(-1, d)[:L]
        r_u = sd["user_embedder.weight"][int(user)] @ sd["user_mask_layer.weight"][0] + sd["user_mask_layer.bias"][0]
        mask = torch.full((L, L), float("-inf"))
        mask[torch.tril(torch.ones(L, L, dtype=torch.bool))] = r_u
        mask[:, L - 1] = 1.0
        mask[:, seq == 0] = float("-inf")
        for l in range(cfg.n_layers):
            p = f"decoder.layers.{l}."
            qkv = F.linear(x, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"])
            q, k, v = (t.reshape(L, H, hd).transpose(0, 1) for t in qkv.split(d, dim=1))
            att = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(hd) + mask, dim=-1)
            o = (att @ v).transpose(0, 1).reshape(L, d)
            x = F.layer_norm(x + F.linear(o, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"]), (d,),
                             sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
            x = F.layer_norm(x + self.c[l], (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
            f = F.linear(F.relu(F.linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])), sd[p + "linear2.weight"],
                         sd[p + "linear2.bias"])
            x = F.layer_norm(x + f, (d,), sd[p + "norm3.weight"], sd[p + "norm3.bias"], 1e-5)
        return x

    def path_step_like_reference(self, seq, user, hep):
        """One path-search step the way the reference computes it (influentialRS.py:414-450): decoder, logits of ALL rows,
        softmax over the catalog for every row, top-100 of row `hep`, first candidate outside the window, shift.
        Returns (next item id, updated window)."""
        x = self.decode(seq, user)
        p = torch.softmax(F.linear(x, self.sd["project.weight"], self.sd["project.bias"]), dim=1)
        vals, ids = torch.topk(p[hep], 100)
        ids = ids + 1
        keep = ~torch.isin(ids, seq[:hep + 1])
        nxt = int(ids[keep][0])
        new = seq.clone()
        new[:-2] = seq[1:-1]
        new[-2] = nxt
        return nxt, new
