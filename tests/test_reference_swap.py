"""not-gpu, build container only (skipped where /root/reference is absent): the
reference's UNMODIFIED pipeline.py imports and drives the drop-in classes after
the import swap of INTEGRATION.md section 1."""
import argparse
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present on this box")


def test_reference_pipeline_load_model_builds_dropin_classes(tmp_path, monkeypatch):
    import influentialrs_amd.model as amd_model
    from influentialrs_amd import synth
    from influentialrs_amd.model import evaluator, influentialRS, layers, uRS
    monkeypatch.syspath_prepend(REF)
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    if not hasattr(np, "Inf"):
        monkeypatch.setattr(np, "Inf", np.inf, raising=False)
    for name, mod in (("wandb", types.ModuleType("wandb")), ("model", amd_model), ("model.influentialRS", influentialRS),
                      ("model.uRS", uRS), ("model.evaluator", evaluator), ("model.layers", layers)):
        monkeypatch.setitem(sys.modules, name, mod)
    for m in ("pipeline", "evaluator_pipeline", "data_provider", "utils"):
        monkeypatch.delitem(sys.modules, m, raising=False)
    monkeypatch.chdir(tmp_path)
    import pipeline  # the reference's file, unmodified
    assert pipeline.IRSNN is influentialRS.IRSNN and pipeline.InfluentialNet is influentialRS.InfluentialNet

    cfg = synth.make_config("tiny")
    cfg.model_store_path = str(tmp_path) + "/"
    cfg.dataset = "syn"
    src = influentialRS.InfluentialNet(cfg)
    src.load_state_dict({k: torch.from_numpy(v) for k, v in synth.irn_state_dict(cfg, 5).items()})
    handler = influentialRS.IRSNN(cfg, src, "cpu")
    os.makedirs(tmp_path / "syn")
    # checkpoint dict exactly as pipeline.train_model writes it (pipeline.py:110-114)
    torch.save({"epoch": 11, "state_dict": src.state_dict(), "optimizer": handler.optimizer.state_dict()},
               tmp_path / "syn" / "irn_params.pth.tar")
    irn = pipeline.load_model(cfg, device="cpu")  # pipeline.py:129-148
    assert isinstance(irn, influentialRS.IRSNN)
    for k, v in src.state_dict().items():
        assert torch.equal(irn.net.state_dict()[k], v)
    # evaluator side: evaluator_pipeline.load_evaluator (evaluator_pipeline.py:122-145)
    import evaluator_pipeline
    ecfg = synth.make_config("eval_tiny")
    ecfg_ns = argparse.Namespace(**vars(ecfg))
    snet = uRS.SampleNet(ecfg)
    ev = evaluator.Evaluator(ecfg, snet, "cpu")
    torch.save({"epoch": 11, "state_dict": snet.state_dict(), "optimizer": ev.optimizer.state_dict()},
               tmp_path / "syn" / "eval_params.pth.tar")
    cfg.n_item, cfg.n_user = ecfg.n_item, ecfg.n_user
    ev2 = evaluator_pipeline.load_evaluator(cfg, ecfg_ns, device="cpu")
    assert isinstance(ev2, evaluator.Evaluator) and isinstance(ev2.net, uRS.SampleNet)
