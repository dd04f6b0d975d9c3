"""MI355X-native counterpart of the reference's model/uRS.py (SampleNet, the
evaluator's next-item recommender): same constructor, forward/decoding
signatures and state_dict keys (word_embedder, pos_embedder.pe, decoder.*,
project.*).  Standard causal 0/-inf mask + key padding on post-padded input
(reference uRS.py:47-64).  Eval mode runs on the HIP engine; training mode is
stock PyTorch autograd."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .._lib import IRS_MASK_CAUSAL
from ._backend import HipBackend
from .layers import PositionalEncoding


class SampleNet(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.PAD_ID = 0
        self.item_embed_path = None
        self.n_item = config.n_item
        self.max_len = config.max_len
        self.n_layers = config.n_layers
        self.n_heads = config.n_heads
        self.embed_dim = config.emb_dim
        self.ffn_dim = config.ffn_dim
        self.dropout = config.dropout
        self.word_embedder = nn.Embedding(self.n_item + 1, self.embed_dim, padding_idx=self.PAD_ID)
        self.pos_embedder = PositionalEncoding(self.embed_dim, self.max_len)
        self.decoder = nn.TransformerDecoder(
            decoder_layer=nn.TransformerDecoderLayer(d_model=self.embed_dim, nhead=self.n_heads,
                                                     dim_feedforward=self.ffn_dim, dropout=self.dropout,
                                                     activation="relu"),
            num_layers=self.n_layers)
        self.project = nn.Linear(self.embed_dim, self.n_item)
        self._hip = HipBackend(self, IRS_MASK_CAUSAL)

    def load_state_dict(self, state_dict, strict=True, **kw):
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        return super().load_state_dict(sd, strict=strict, **kw)

    def shard_items(self, rank: int, world: int, drop_full: bool = True):
        """Item shard [lo, hi) of the catalog on this GPU; see InfluentialNet.shard_items."""
        self._hip.set_sharding(rank, world, drop_full)

    def _generate_square_subsequent_mask(self, sz):
        mask = (torch.triu(torch.ones(sz, sz)) == 1).transpose(0, 1)
        return mask.float().masked_fill(mask == 0, float("-inf")).masked_fill(mask == 1, float(0.0))

    def _decoding_autograd(self, seq):
        pad = seq.eq(self.PAD_ID)
        enc = torch.zeros(self.max_len, seq.size(0), self.embed_dim, device=seq.device)
        x = self.word_embedder(seq) * math.sqrt(self.embed_dim) + self.pos_embedder(seq)
        x = F.dropout(x, self.dropout, self.training).transpose(0, 1)
        mask = self._generate_square_subsequent_mask(seq.size(1)).to(seq.device)
        padf = torch.zeros_like(pad, dtype=torch.float32).masked_fill(pad, float("-inf"))
        return self.decoder(tgt=x, memory=enc, tgt_mask=mask, tgt_key_padding_mask=padf).transpose(0, 1)

    def _pad_to_len(self, seq):
        """The evaluator feeds L-1 or L tokens (`seqs[:, :-1]`, evaluator.py:56,189);
        the engine's window is max_len wide: right-pad with PAD (post-padded input,
        causal mask: trailing pads do not influence earlier rows)."""
        L = seq.size(1)
        if L == self.max_len:
            return seq.contiguous(), L
        if L > self.max_len:
            raise ValueError(f"sequence length {L} exceeds max_len {self.max_len}")
        return F.pad(seq, (0, self.max_len - L), value=self.PAD_ID).contiguous(), L

    def decoding(self, dec_inp_seq):
        if self.training:
            return self._decoding_autograd(dec_inp_seq)
        seq, L = self._pad_to_len(dec_inp_seq)
        eng = self._hip.get(seq.size(0), 1)
        x, _, _ = eng.decode(seq, None, want_x=True)
        return x[:, :L]

    def forward(self, dec_inp_seq):
        if self.training:
            return self.project(self._decoding_autograd(dec_inp_seq))
        B, L = dec_inp_seq.shape
        x = self.decoding(dec_inp_seq).contiguous()
        eng = self._hip.get(B, B * L)
        if eng.world != 1:
            raise RuntimeError("forward() materialises [B, L, n_item]; with item sharding use the Evaluator handlers")
        return eng.score_dense(x.view(B * L, self.embed_dim)).view(B, L, self.n_item)

    def decode_rows(self, seqs, rows_b, rows_pos):
        """x[rows_b[i], rows_pos[i], :] for a list of (sequence, position) pairs."""
        seq, L = self._pad_to_len(seqs)
        eng = self._hip.get(seq.size(0), max(int(rows_b.numel()), 1))
        x, _, _ = eng.decode(seq, None, want_x=True)
        return x[rows_b, rows_pos].contiguous()
