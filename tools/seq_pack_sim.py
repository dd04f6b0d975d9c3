#!/usr/bin/env python3
"""Lab: what a SEQUENCE-RESIDENT decoder workgroup would pay in padding ("sequence-resident decoder").

Today the decoder kernels run on globally packed non-pad tokens (tiles of 32 tokens that may span sequences: no padding at
all).  A kernel that keeps a whole sequence's K / V on chip needs every sequence inside ONE workgroup, token tiles aligned to
the sequence, and the workgroup's wave slots filled with whole sequences.  This script draws the bench's own window lengths
(bench.gpu_windows: history log-normal, median 95, sigma 0.95, clipped to [18, L - 1], + the target) and reports the lane
efficiency (useful token rows / rows a workgroup grid would execute) of the packings such a kernel could use.

usage: python tools/seq_pack_sim.py [users=4096] [L=200] [waves=8]"""
import sys
import numpy as np

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
NW = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rng = np.random.default_rng(0)
hl = np.clip(np.exp(rng.standard_normal(B) * 0.95 + np.log(95.0)), 18, 2276).astype(int).clip(max=L - 1)
n = hl + 1
tok = int(n.sum())
print(f"{B} users, L = {L}: mean {n.mean():.1f} tokens per window, packed row fraction {tok / (B * L):.3f}")
nb = (n + 15) // 16          # 16-token blocks (the attention kernels' query blocks)
T = (nb + 1) // 2            # 32-token tiles = one wave's tokens in the fused layer kernel
print("windows per tile count 1..%d:" % T.max(), np.bincount(T)[1:].tolist())


def in_order(items, cap):
    bins, cur = 0, 0
    for t in items:
        if cur + t > cap:
            bins, cur = bins + 1, 0
        cur += t
    return bins + (cur > 0)


def ffd(items, cap):
    free = []
    for t in sorted(items, reverse=True):
        for i in range(len(free)):
            if free[i] >= t:
                free[i] -= t
                break
        else:
            free.append(cap - t)
    return len(free)


rows = lambda bins, per: bins * per
print(f"tiles aligned to the sequence, no workgroup constraint:   {tok / (T.sum() * 32):.3f}")
for name, f in (("in window order", in_order), ("first-fit decreasing (needs a sort)", ffd)):
    b = f(T.tolist(), NW)
    print(f"{NW} waves x 32 tokens, whole sequences, {name:38s}: {b} workgroups, lane efficiency {tok / rows(b, NW * 32):.3f}")
    b = f(nb.tolist(), 2 * NW)
    print(f"{NW} waves x 2 x 16-token blocks (a wave's halves from two sequences), {name:22s}: {b} workgroups, lane efficiency {tok / rows(b, NW * 32):.3f}")
