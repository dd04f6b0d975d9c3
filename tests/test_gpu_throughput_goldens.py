"""-m gpu: the THROUGHPUT kernels (what bench.py's headline runs: packed rows, fragment-major activations, the fused
layer kernel k_block_x6 / k_block, the packed-sequence attention) pinned DIRECTLY to the reference's goldens.

tests/test_gpu_decoder_path.py runs the goldens' 32 (c2) / 8 (c3) users per call, i.e. 6400 / 1600 token rows: below the
32768-row switch of irs_launch_decode, so those tests exercise the small-batch kernels.  Here the golden users are tiled
(c2 x 6 = 192 windows = 38400 token rows; c3 x 24 = 192 windows; round 5: c4d x 6 -- d = 256, 8 heads, the decoder of
BASELINE configs[3] / [4] on k_block_x6<.., NT = 8>), shuffled, and sent through irs_decode +
irs_score_topk + irs_generate_paths (stream and hipGraph) in ALL decoder arithmetic modes (IRS_GEMM_H3, the default,
IRS_GEMM_X6 and IRS_GEMM_F32): decoder rows, ranked top-100 ids, 20-step paths and the early-success count are compared with what the
unmodified reference produced for those users (reference model/influentialRS.py:412-450, 340-390).

Tolerances: decoder rows 2e-5 (float32 MFMAs) / 4e-5 (split-float16 and split-bf16 MFMAs) absolute on O(1) LayerNorm outputs; top-100
values 5e-5; ranked ids order-exact outside runs of reference scores closer than TAU (rank_check.py), with the set of
users that are NOT id-for-id identical asserted exactly (NEAR_TIE_USERS); paths and early successes exact."""
import numpy as np
import pytest
import torch

from influentialrs_amd import synth
from influentialrs_amd._lib import IRS_GEMM_F32, IRS_GEMM_H3, IRS_GEMM_X6, IRS_SWEEP_BF16
from gpu_util import make_engine
from parity_record import check_exact, record
from rank_check import check_ranked

pytestmark = pytest.mark.gpu

X_TOL = {IRS_GEMM_F32: 2e-5, IRS_GEMM_X6: 4e-5, IRS_GEMM_H3: 4e-5}
TAU = 2e-5
MODES = {"h3": IRS_GEMM_H3, "x6": IRS_GEMM_X6, "f32": IRS_GEMM_F32}
# golden users whose top-100 ids differ from the reference's inside a run of reference gaps < TAU (everything else is
# identical id for id); filled from a recording run (IRS_RECORD_PARITY=1), see profiles/r04/parity_counts.json
NEAR_TIE_USERS = {
    # (round 5: with the weight planes out of the float16 subnormal range h3 lands on the oracle's side of c2 user 20's 2.4e-7
    #  gap, like x6, the small-batch kernels and the CPU oracle; the float32-MFMA kernels on the reference's side)
    ("irn_c2", "h3"): [20], ("irn_c2", "x6"): [20], ("irn_c2", "f32"): [],
    ("irn_c3", "h3"): [], ("irn_c3", "x6"): [], ("irn_c3", "f32"): [],
    ("irn_c4d", "h3"): [], ("irn_c4d", "x6"): [], ("irn_c4d", "f32"): [],
}
# d = 256 (irn_c4d, round 5): K = 256 contractions accumulate twice as long on the truncating 16-bit pipe
# (observed against the reference's rows, profiles/r05/parity_counts.json: f32 1.1e-5, x6 1.7e-5, h3 2.6e-5)
X_TOL_D256 = {IRS_GEMM_F32: 2e-5, IRS_GEMM_X6: 4e-5, IRS_GEMM_H3: 4e-5}
_ENG = {}


def _engine(cfgname, B):
    if cfgname not in _ENG:
        _ENG.clear()  # one catalog resident at a time (c3: 1M x 128)
        cfg = synth.make_config(cfgname)
        sd = synth.irn_state_dict(cfg, 1234)
        _ENG[cfgname] = (cfg, make_engine(cfg, sd, max_rows=B, max_seqs=B))
    return _ENG[cfgname]


def _trim_after_target(paths, targets):
    out, n = paths.copy(), 0
    for i in range(out.shape[0]):
        pos = np.where(out[i] == targets[i])[0]
        if len(pos):
            n += 1
            out[i, pos[0] + 1:] = 0
    return out, n


@pytest.mark.parametrize("name,cfgname,reps", [("irn_c2", "c2", 6), ("irn_c3", "c3", 24), ("irn_c4d", "c4d", 6)])
@pytest.mark.parametrize("mode", ["h3", "x6", "f32"])
def test_throughput_kernels_reproduce_reference_goldens(golden, name, cfgname, reps, mode):
    _run(golden, name, cfgname, reps, mode, False)


@pytest.mark.parametrize("name,cfgname,reps", [("irn_c2", "c2", 6), ("irn_c3", "c3", 24)])
def test_sequence_resident_layers_reproduce_reference_goldens(golden, name, cfgname, reps):
    """Round 5: the sequence-resident layer kernel (irs_set_decoder_seq: q | k | v, attention and the layer body of whole
    sequences in ONE launch per layer, K / V in LDS) on the same tiled golden users: rows, ranked ids, paths and early successes
    against the unmodified reference's, with the float16-plane arithmetic's near-tie sets."""
    _run(golden, name, cfgname, reps, "h3", True)
    # (the rows-only decodes above did take the sequence-resident kernel: _run asserts decoder_seq_last)


def _run(golden, name, cfgname, reps, mode, seq_resident):
    g = golden(name)
    B0, L = g["seqs"].shape
    B = B0 * reps
    assert B * L > 32768, "must be above the switch to the throughput kernels"
    cfg, eng = _engine(cfgname, B)
    eng.decoder_gemm = MODES[mode]
    eng.decoder_seq = seq_resident
    assert eng.decoder_seq == seq_resident
    tol = (X_TOL_D256 if cfg.emb_dim == 256 else X_TOL)[MODES[mode]]
    try:
        rng = np.random.default_rng(20261004)
        src = rng.permutation(np.tile(np.arange(B0), reps))
        seqs, users, targets = g["seqs"][src], g["users"][src], g["targets"][src]
        seq, usr = torch.from_numpy(seqs).cuda(), torch.from_numpy(users).cuda()
        pos = torch.full((B,), L - 2, dtype=torch.int32, device="cuda")
        # --- decoder rows: the consumed row of every window against the reference's (rows-only decode: packed tokens,
        #     k | v-only tail in front of the last layer) and the full decode (every row computed) against it too
        _, xr, ru = eng.decode(seq, usr, want_x=False, pos=pos, want_r_u=True)
        assert eng.decoder_seq_last == bool(seq_resident)
        x_full, xr_full, _ = eng.decode(seq, usr, want_x=True, pos=pos)
        xr_h, ru_h = xr.cpu().numpy(), ru.cpu().numpy()
        assert np.abs(ru_h - g["r_u"][src]).max() < 1e-6
        err = np.abs(xr_h - g["x_hep"][src]).max()
        assert err < tol, (name, mode, err)
        err_full = np.abs(xr_full.cpu().numpy() - g["x_hep"][src]).max()
        assert err_full < tol, (name, mode, err_full)
        record(f"throughput{'_seq' if seq_resident else ''}_rows/{name}/{mode}", B0, B0, [], {"max_abs_row_error_vs_reference": float(err), "full_decode": float(err_full)})
        # copies of one user decode to the same bits wherever they sit in the batch
        first = np.array([np.nonzero(src == u)[0][0] for u in range(B0)])
        assert np.array_equal(xr_h, xr_h[first][src])
        # --- ranked ids against the reference's own ranking
        val, ids, st = eng.score_topk(xr, 100, IRS_SWEEP_BF16)
        val, ids = val.cpu().numpy(), ids.cpu().numpy()
        assert not (st.cpu().numpy() & 1).any()
        near = set()
        for i in range(B):
            u = int(src[i])
            assert np.abs(val[i] - g["top_vals"][u][:100]).max() < 5e-5
            if not check_ranked(ids[i], g["top_ids0"][u], g["top_gaps"][u], TAU):
                near.add(u)
        check_exact(f"throughput{'_seq' if seq_resident else ''}/{name}/{mode}", near, NEAR_TIE_USERS[(name, mode)], B0)
        # --- 20-step (c3: 4-step) greedy paths, stream launches and the captured step
        P = int(g["meta"][2])
        for use_graph in (False, True):
            work = seq.clone()
            hep = pos.clone()
            paths, st2 = eng.generate_paths(work, usr, hep, P, k=100, sweep=IRS_SWEEP_BF16, use_graph=use_graph)
            torch.cuda.synchronize()
            assert (st2.cpu().numpy() & 2).sum() == 0
            trimmed, n_early = _trim_after_target(paths.cpu().numpy(), targets)
            assert np.array_equal(trimmed, g["paths"][src]), (name, mode, use_graph)
            assert n_early == reps * int(g["n_early_success"]) and n_early > 0
    finally:
        eng.decoder_gemm = IRS_GEMM_H3
        eng.decoder_seq = None  # auto
