#!/usr/bin/env python3
"""Lab: where a wave of the sequence-resident decoder launch spends its cycles.  Needs a stamped build of the library
(tools/seq_lab.sh stamp -> tools/seqlab_stamp.so: -DIRS_LAB -DX6_STAMP=2, s_memtime at the phase boundaries of
k_block_x6<.., SEQ>, summed over the layers of the launch).  usage: IRS_LAB_LIB=tools/seqlab_stamp.so python tools/seq_stamps.py [users=4096]"""
import ctypes, os, sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np, torch
from influentialrs_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ["IRS_LAB_LIB"])
import bench
from influentialrs_amd import synth
from gpu_util import make_engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
cfg = synth.make_config("c2")
eng = make_engine(cfg, synth.irn_state_dict(cfg, 1234), max_rows=B, max_seqs=B)
seqs = bench.gpu_windows(B, cfg.max_len, cfg.n_item, dev, seed=3)
if len(sys.argv) > 2:  # every window cut to the same number of tokens (history + target): no imbalance between the waves
    n = int(sys.argv[2])
    full = seqs.clone()
    full[full == 0] = 7
    col = torch.arange(cfg.max_len, device=dev)[None, :]
    seqs = torch.where(col >= cfg.max_len - n, full, torch.zeros_like(full))
users = torch.randint(0, cfg.n_user, (B,), device=dev, generator=torch.Generator(device=dev).manual_seed(1))
pos = torch.full((B,), cfg.max_len - 2, dtype=torch.int32, device=dev)
eng.decoder_seq = True
for _ in range(3):
    eng.decode(seqs, users, want_x=False, pos=pos)
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.decode(seqs, users, want_x=False, pos=pos)
torch.cuda.synchronize()
print("decode (stamped build) %.3f ms, %d users" % ((time.perf_counter() - t0) * 1e3, B))
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.irs_lab_seq_stamps.restype = ctypes.c_void_p
ptr = lib.irs_lab_seq_stamps()
nwg = int(eng.debug_buffer(6, 1, torch.int32)[0].item())
n = nwg * 8 * 24
buf = torch.empty(n, dtype=torch.int64, device=dev)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(n * 8), ctypes.c_int(3))
st = buf.cpu().numpy().reshape(nwg, 8, 24).astype(np.float64)
tot = st[:, :, 0]
names = ["front: epilogues (bias, q scratch, K / V plane split, image writes)", "front: barrier (images complete)", "front: attention blocks", "fence + tile / residual loads",
         "out-projection steps", "LayerNorm 1 / 2", "planes + feed-forward steps", "LayerNorm 3 + planes", "empty step (next layer's vectors)", "front: q|k|v steps (3 one-tile steps per head)"]
print("workgroups %d; cycles per wave (s_memtime, 100 MHz): mean %.0f  min %.0f  max %.0f; per workgroup max: mean %.0f" %
      (nwg, tot.mean(), tot.min(), tot.max(), tot.max(1).mean()))
ph = st[:, :, 8:8 + len(names)]
acc = ph.sum(2)
print("phases cover %.1f %% of a wave's life" % (100 * acc.mean() / tot.mean()))
for i, nm in enumerate(names):
    print("  %-72s %6.1f %%   (mean %.0f ticks)" % (nm, 100 * ph[:, :, i].mean() / tot.mean(), ph[:, :, i].mean()))
if st[:, :, 2].sum() > 0:  # (X6_STAMP=1 builds: the per-step stamps too)
    for i, nm in ((1, "all steps: wait for the next step's DMA (vmcnt)"), (2, "all steps: mid-step barrier"), (3, "all steps: DMA issue"),
                  (18, "  of which: first barrier behind a head's attention"), (19, "  of which: first barrier of the layer body")):
        print("  %-72s %6.1f %%   (mean %.0f ticks)" % (nm, 100 * st[:, :, i].mean() / tot.mean(), st[:, :, i].mean()))
for i, nm in () if st[:, :, 20:24].sum() == 0 else ((20, "  attention blocks: q read + split, score tiles, masks"), (21, "  attention blocks: target column"), (22, "  attention blocks: max, exp, P split, V reads, P.V"),
              (23, "  attention blocks: normalise + store")):
    print("  %-72s %6.1f %%   (mean %.0f ticks)" % (nm, 100 * st[:, :, i].mean() / tot.mean(), st[:, :, i].mean()))
# spread between the workgroups: the launch ends with its slowest workgroup
wg = tot.max(1)
print("workgroup life: p5 %.0f  p50 %.0f  p95 %.0f  max %.0f ticks; launch / (sum of workgroup lives / 256 CUs) = imbalance" %
      tuple(np.percentile(wg, [5, 50, 95, 100])))
print("sum of workgroup lives / 256 = %.0f ticks = %.3f ms" % (wg.sum() / 256, wg.sum() / 256 / 1e5))
