#!/usr/bin/env python3
"""B-ref: wall time of the UNMODIFIED reference's pipeline.test_model
(/root/reference/pipeline.py:151-247) on synthetic ml-1m-shaped inputs, batch size 1 (the only batch size the
published IRN runs at), all host threads.  Container-only: the reference never travels to the GPU box.
It calibrates the B-equiv leg of bench.py's cpu_baseline (the CPU restatement doing what the reference does)
against the real thing.  Same harness-side shims as tests/golden/make_golden.py (torch / numpy API drift only).

    python tools/time_reference.py [c1 c2 default] [--users 16]
"""
import argparse
import contextlib
import io
import os
import re
import sys
import tempfile
import time
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import numpy as np
import torch
from torch.optim import lr_scheduler

if not hasattr(np, "Inf"):
    np.Inf = np.inf
sys.modules.setdefault("wandb", types.ModuleType("wandb"))
_RLROP = lr_scheduler.ReduceLROnPlateau


class _RLROPCompat(_RLROP):
    def __init__(self, *a, verbose=None, **k):
        super().__init__(*a, **k)


lr_scheduler.ReduceLROnPlateau = _RLROPCompat

from influentialrs_amd import synth  # noqa: E402


def run(cfg_name, n_users):
    import pipeline  # reference, unmodified
    from model.influentialRS import InfluentialNet, IRSNN
    tmp = tempfile.mkdtemp(dir="/tmp")
    os.chdir(tmp)
    cfg = synth.make_config(cfg_name)
    for k, v in dict(model_store_path=tmp + "/", dataset="syn", method="IRN", use_train=False, gap_len=0, batch_size=1,
                     top_k=20, use_h=True, max_path_len=20, sample=False, sample_k=3, use_wandb=False).items():
        setattr(cfg, k, v)
    os.makedirs(os.path.join(tmp, "syn"))
    torch.manual_seed(0)
    net = InfluentialNet(cfg)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.irn_state_dict(cfg, 1234).items()})
    irn = IRSNN(cfg, net, "cpu")
    torch.save({"epoch": 3, "state_dict": net.state_dict(), "optimizer": irn.optimizer.state_dict()},
               os.path.join(tmp, "syn", "irn_params.pth.tar"))
    hists = synth.user_histories(n_users, cfg.n_item, seed=7)
    rows = synth.eval_rows(hists, cfg.n_item, seed=11)

    class _DP:
        def get_random_evaluate_data(self, **kw):
            return [list(r) for r in rows]

    orig_save = np.save

    def save_ragged(file, arr, *a, **k):
        try:
            return orig_save(file, arr, *a, **k)
        except ValueError:
            o = np.empty(len(arr), dtype=object)
            for i, x in enumerate(arr):
                o[i] = x
            return orig_save(file, o, *a, **k)

    pipeline.np.save = save_ragged
    buf = io.StringIO()
    t0 = time.perf_counter()
    try:
        with contextlib.redirect_stdout(buf):
            pipeline.test_model(cfg, _DP(), device="cpu")
    finally:
        pipeline.np.save = orig_save
    dt = time.perf_counter() - t0
    # per user: 1 PIF pass + 1 ranking forward + 20 path steps; consumed rows = 21 per user
    per_user = dt / n_users
    print(f"{cfg_name}: {n_users} users in {dt:.2f}s = {per_user * 1e3:.1f} ms/user = {per_user / 22 * 1e3:.2f} ms per forward "
          f"(22 per user) -> {21 * cfg.n_item / per_user:.3e} consumed pairs/s; threads={torch.get_num_threads()} "
          f"torch={torch.__version__}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cfgs", nargs="*", default=["c1", "c2"])
    ap.add_argument("--users", type=int, default=16)
    a = ap.parse_args()
    for c in a.cfgs:
        run(c, a.users)
