"""-m gpu: beam search (build-defined extension, SURVEY row A9: no reference oracle;
the CPU restatement oracle_np.beam_search is the yard-stick).  beam = 1 must equal the
greedy search (and hence the reference goldens) id for id."""
import numpy as np
import pytest
import torch

from influentialrs_amd import synth
from influentialrs_amd._lib import IRS_SWEEP_BF16, IRS_SWEEP_F32
from influentialrs_amd.model.influentialRS import IRSNN, InfluentialNet
from gpu_util import make_engine

pytestmark = pytest.mark.gpu


def _setup(cfgname, n):
    cfg = synth.make_config(cfgname)
    sd = synth.irn_state_dict(cfg, 1234)
    hists = synth.user_histories(max(n, 8), cfg.n_item, seed=7)
    rows = synth.eval_rows(hists, cfg.n_item, seed=11)[:n]
    raws, seqs, users, targets, labels = synth.collate_eval_irs(rows, cfg.max_len, gap_len=0)
    return cfg, sd, seqs, users, targets


@pytest.mark.parametrize("gname,cfgname", [("irn_default", "default"), ("irn_c4d", "c4d")])
@pytest.mark.parametrize("use_graph", [False, True])
def test_beam1_equals_greedy_and_reference(golden, use_graph, gname, cfgname):
    """(irn_c4d, round 5: C5's decoder shape -- the beam loop's decodes of 32 windows run k_block_small_wide<256>.)"""
    g = golden(gname)
    cfg = synth.make_config(cfgname)
    sd = synth.irn_state_dict(cfg, 1234)
    B, L = g["seqs"].shape
    eng = make_engine(cfg, sd, max_rows=B, max_seqs=B)
    hep = torch.full((B,), L - 2, dtype=torch.int32, device="cuda")
    P = int(g["meta"][2])
    paths, scores, st = eng.beam_search(torch.from_numpy(g["seqs"]).cuda(), torch.from_numpy(g["users"]).cuda(), hep, P, 1,
                                        use_graph=use_graph)
    got = paths[:, 0].cpu().numpy()
    for i in range(B):  # the reference zeroes the tail after the target's first occurrence (influentialRS.py:459-467)
        pos = np.where(got[i] == g["targets"][i])[0]
        if len(pos):
            got[i, pos[0] + 1:] = 0
    assert np.array_equal(got, g["paths"])


@pytest.mark.parametrize("cfgname,beam,P,use_graph", [("tiny", 4, 7, False), ("tiny", 4, 6, True), ("tiny", 32, 5, False),
                                                      ("default", 3, 5, True)])
def test_beam_matches_cpu_restatement(oracle, cfgname, beam, P, use_graph):
    cfg, sd, seqs, users, targets = _setup(cfgname, 3)
    B, L = seqs.shape
    eng = make_engine(cfg, sd, max_rows=B * beam, max_seqs=B * beam)
    hep = torch.full((B,), L - 2, dtype=torch.int32, device="cuda")
    paths, scores, st, fin = eng.beam_search(torch.from_numpy(seqs).cuda(), torch.from_numpy(users).cuda(), hep, P, beam,
                                             sweep=IRS_SWEEP_F32, use_graph=use_graph, want_windows=True)
    paths, scores = paths.cpu().numpy(), scores.cpu().numpy()
    op, osc = oracle.beam_search(sd, cfg, seqs, users, max_path_len=P, gap_len=0, beam=beam)
    assert np.allclose(scores, osc, rtol=0, atol=2e-4), (scores, osc)
    # ids exact wherever consecutive beam scores are separated by more than the float32 LSE noise
    for b in range(B):
        gaps = np.abs(np.diff(osc[b]))
        n_safe = beam if len(gaps) == 0 or gaps.min() > 1e-4 else int(np.argmax(gaps <= 1e-4)) + 1
        assert np.array_equal(paths[b, :n_safe], op[b, :n_safe]), (b, paths[b], op[b])
    if beam == 1:  # the greedy entry point (irs_generate_paths) walks the same paths
        seq2 = torch.from_numpy(seqs).cuda()
        hep2 = torch.full((B,), L - 2, dtype=torch.int32, device="cuda")
        gp = eng.generate_paths(seq2, torch.from_numpy(users).cuda(), hep2, P, use_graph=not graph)
        gp = (gp[0] if isinstance(gp, tuple) else gp).cpu().numpy()
        assert np.array_equal(gp, paths[:, 0])
    # scores are sorted, windows end with the path
    assert (np.diff(scores, axis=1) <= 0).all()
    w = fin.cpu().numpy()
    n = min(P, L - 1)
    assert np.array_equal(w[:, :, L - 1 - n:L - 1], paths[:, :, P - n:].astype(np.int64))


def test_beam_frontend_kwarg():
    cfg, sd, seqs, users, targets = _setup("tiny", 3)
    net = InfluentialNet(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.to("cuda:0")
    irn = IRSNN(cfg, net, "cuda:0").eval()
    seq, u, t = (torch.from_numpy(a).cuda() for a in (seqs, users, targets))
    with torch.no_grad():
        p1, _, _, _ = irn.get_seq_in_batch(seq, u, t, 6, 0)
        pb1, _, _, _ = irn.get_seq_in_batch(seq, u, t, 6, 0, beam_width=1)
        pb4, _, hist, _ = irn.get_seq_in_batch(seq, u, t, 6, 0, beam_width=4)
    assert np.array_equal(p1, pb1)
    allp, alls = irn.last_beams
    assert allp.shape == (3, 4, 6) and np.array_equal(allp[:, 0], pb4)
    # the best beam's cumulative log-probability is at least the greedy path's
    assert pb4.shape == p1.shape and len(hist) == 3


def _random_beam_cases(n, seed):
    g = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        hd = int(g.choice([8, 16, 32]))
        H = int(g.integers(1, 5))
        out.append(dict(emb_dim=hd * H, n_heads=H, ffn_dim=int(g.choice([32, 64, 100])), max_len=int(g.choice([10, 24, 40])),
                        n_layers=int(g.integers(1, 3)), n_item=int(g.choice([150, 900, 5000])), n_user=9,
                        B=int(g.integers(1, 5)), beam=int(g.integers(1, 9)), P=int(g.integers(2, 8)), graph=bool(g.integers(0, 2))))
    return out


@pytest.mark.parametrize("case", _random_beam_cases(int(__import__("os").environ.get("IRS_RANDOM_SHAPES", "10")),
                                                    int(__import__("os").environ.get("IRS_RANDOM_SHAPES_SEED", "5"))))
def test_beam_random_cases_match_cpu_restatement(oracle, case):
    """Randomly drawn model shapes, beam widths, path lengths and batch sizes (stream launches and the captured graph)
    against oracle_np.beam_search: scores within the float32 log-sum-exp noise, paths id for id wherever the beams
    are separated by more than that noise."""
    c = dict(case)
    B, beam, P, graph = c.pop("B"), c.pop("beam"), c.pop("P"), c.pop("graph")
    cfg = synth.make_config("tiny", **c)
    sd = synth.irn_state_dict(cfg, 77)
    hists = synth.user_histories(8, cfg.n_item, seed=7)
    rows = synth.eval_rows(hists, cfg.n_item, seed=11)[:B]
    raws, seqs, users, targets, labels = synth.collate_eval_irs(rows, cfg.max_len, gap_len=0)
    L = cfg.max_len
    eng = make_engine(cfg, sd, max_rows=B * beam, max_seqs=B * beam)
    hep = torch.full((B,), L - 2, dtype=torch.int32, device="cuda")
    paths, scores, st = eng.beam_search(torch.from_numpy(seqs).cuda(), torch.from_numpy(users).cuda(), hep, P, beam,
                                        sweep=IRS_SWEEP_BF16, use_graph=graph)
    paths, scores = paths.cpu().numpy(), scores.cpu().numpy()
    op, osc = oracle.beam_search(sd, cfg, seqs, users, max_path_len=P, gap_len=0, beam=beam)
    fin = np.isfinite(osc)
    assert np.array_equal(fin, np.isfinite(scores))
    assert np.allclose(scores[fin], osc[fin], rtol=0, atol=2e-4), (scores, osc)
    for b in range(B):
        nb = int(fin[b].sum())
        gaps = np.abs(np.diff(osc[b, :nb]))
        n_safe = nb if len(gaps) == 0 or gaps.min() > 1e-4 else int(np.argmax(gaps <= 1e-4)) + 1
        assert np.array_equal(paths[b, :n_safe], op[b, :n_safe]), (b, paths[b], op[b])
    if beam == 1:  # the greedy entry point (irs_generate_paths) walks the same paths
        seq2 = torch.from_numpy(seqs).cuda()
        hep2 = torch.full((B,), L - 2, dtype=torch.int32, device="cuda")
        gp = eng.generate_paths(seq2, torch.from_numpy(users).cuda(), hep2, P, use_graph=not graph)
        gp = (gp[0] if isinstance(gp, tuple) else gp).cpu().numpy()
        assert np.array_equal(gp, paths[:, 0])
