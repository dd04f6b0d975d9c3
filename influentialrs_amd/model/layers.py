"""Counterpart of the reference's model/layers.py (same names and semantics)."""
import numpy as np
import torch
import torch.nn as nn


class PositionalEncoding(nn.Module):
    """Sinusoidal table registered as buffer `pe` [1, max_len, d_model]; forward
    returns pe[:, :x.size(1)] (reference model/layers.py:17-32)."""

    def __init__(self, d_model, max_len):
        super().__init__()
        from ..synth import positional_encoding  # the package's one definition of the table (float32, [1, max_len, d])
        self.register_buffer("pe", torch.from_numpy(positional_encoding(d_model, max_len).copy()))

    def forward(self, x):
        return self.pe[:, :x.size(1)]


def get_end_index(seq, pad=0):
    """End position of a post-padded sequence (reference model/layers.py:34-42)."""
    pos = np.where(seq == pad)[0]
    return len(seq) - 1 if len(pos) == 0 else pos[0] - 1


def get_start_index(seq, pad=0):
    """Start position of a pre-padded sequence (reference model/layers.py:44-49)."""
    return np.where(seq != pad)[0][0]


def get_item_index(seq, item):
    """First position of `item` in `seq`, or -1 (reference model/layers.py:51-59)."""
    pos = np.where(seq == item)[0]
    return -1 if len(pos) == 0 else pos[0]
