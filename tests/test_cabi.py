"""not-gpu: the C-ABI library loads, exports every symbol include/irs_hip.h
declares, and validates arguments before touching a device."""
import ctypes
import os
import re

import pytest

from influentialrs_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(REPO, "include", "irs_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(irs_[a-z_0-9]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/irs_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    assert lib.irs_abi_version() == 1


def test_create_validates_without_a_device():
    lib = _lib.load()
    h = ctypes.c_void_p()

    def mk(**kw):
        base = dict(n_item=1000, n_user=10, d=64, max_len=50, n_heads=4, ffn_dim=256, n_layers=6, u_dim=10,
                    mask_mode=0, max_rows=8, max_k=100, max_seqs=0)
        base.update(kw)
        return _lib.IrsDims(**base)

    for bad in (dict(d=63), dict(d=512), dict(max_len=300), dict(n_heads=3), dict(n_layers=0), dict(max_k=0),
                dict(mask_mode=7), dict(u_dim=0), dict(n_item=0)):
        assert lib.irs_create(ctypes.byref(h), ctypes.byref(mk(**bad)), None) == -1, bad
        assert lib.irs_last_error(None)
    sh = _lib.IrsShard(0, 2, 500, 400)
    assert lib.irs_create(ctypes.byref(h), ctypes.byref(mk()), ctypes.byref(sh)) == -1
    assert lib.irs_create(ctypes.byref(h), ctypes.byref(mk()), None) == 0
    try:
        assert lib.irs_workspace_bytes(h) > 0 and lib.irs_derived_bytes(h) > 0
        fake = ctypes.c_void_p(0x1000)
        assert lib.irs_bind_weight(h, b"project.bias", fake, 1000) == 0
        assert lib.irs_bind_weight(h, b"module.project.bias", fake, 1000) == 0  # DataParallel prefix
        assert lib.irs_bind_weight(h, b"project.bias", fake, 999) == -1
        assert b"expected 1000" in lib.irs_last_error(h)
        assert lib.irs_bind_weight(h, b"decoder.layers.5.norm3.weight", fake, 64) == 0
        assert lib.irs_bind_weight(h, b"decoder.layers.6.norm3.weight", fake, 64) == -1
        assert lib.irs_bind_weight(h, b"nonsense", fake, 1) == -1
        # nothing runs before weights are finalized and a workspace is bound
        assert lib.irs_score_topk(h, fake, 1, 10, 0, fake, fake, fake, None) == -2
    finally:
        lib.irs_destroy(h)


def test_product_has_no_cpu_path():
    import torch
    from influentialrs_amd.engine import Engine, IrsError
    with pytest.raises(IrsError):
        Engine(n_item=100, n_user=4, d=16, max_len=8, n_heads=2, ffn_dim=16, n_layers=1, u_dim=4, mask_mode=0,
               device=torch.device("cpu"))
    # and the product never imports the oracle
    for root, _, files in os.walk(os.path.join(REPO, "influentialrs_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f


def test_product_build_defines_no_lab_switch():
    """The kernels' measurement hooks (X6_NO_MFMA, X6_DUMP, ATTN_STAMP ...) change what a kernel computes or writes; they
    are legal only under IRS_LAB (csrc/irs_internal.h #errors otherwise) and the product build defines neither."""
    from influentialrs_amd import build
    assert not [f for f in build.FLAGS if f.startswith("-D")], build.FLAGS
    with open(os.path.join(REPO, "influentialrs_amd", "csrc", "irs_internal.h")) as fh:
        txt = fh.read()
    assert "#if !defined(IRS_LAB)" in txt and "#error" in txt
    import subprocess
    src = os.path.join(REPO, "influentialrs_amd", "csrc", "path.hip")
    r = subprocess.run([build.HIPCC, "--offload-arch=gfx950", "-std=c++17", "-DX6_NO_MFMA", "-fsyntax-only", src],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "IRS_LAB" in r.stderr
