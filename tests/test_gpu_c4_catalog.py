"""-m gpu: the scoring path at BASELINE.json config 4's FULL size -- 10,000,000 items x d = 256 -- against the
CPU oracle ("restatement as oracle at 10M").  The catalog is drawn on the GPU (10 GB of float32)
and copied to the host once; the oracle's OpenMP chain then scores the same rows over the same bytes.
Bar: bit-exact ids AND values (the fixed-order float32 chain), exact rank counts, 1e-5 relative on log-sum-exp
(a float32 sum of 10^7 terms in a different association order)."""
import numpy as np
import pytest
import torch

from influentialrs_amd import synth
from influentialrs_amd._lib import IRS_SWEEP_BF16, IRS_SWEEP_EXHAUSTIVE, IRS_SWEEP_F32
from influentialrs_amd.engine import Engine
from influentialrs_amd._lib import IRS_MASK_IRN

pytestmark = pytest.mark.gpu

N_ITEM, D, M, K = 10_000_000, 256, 6, 100


@pytest.fixture(scope="module")
def catalog():
    import psutil
    if psutil.virtual_memory().available < 28 * 2 ** 30:
        pytest.skip("needs ~24 GB of host memory for the oracle's copy of the 10M x 256 catalog")
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(20240)
    W = (torch.rand((N_ITEM, D), generator=g, device=dev, dtype=torch.float32) * 2 - 1) / 16.0  # U(+-1/sqrt(d))
    b = torch.randn((N_ITEM,), generator=g, device=dev, dtype=torch.float32) * 0.1
    x = torch.randn((M, D), generator=g, device=dev, dtype=torch.float32)
    Wh = W.cpu().numpy()
    bh = b.cpu().numpy()
    xh = x.cpu().numpy()
    return W, b, x, Wh, bh, xh


@pytest.fixture(scope="module")
def oracle_rows(oracle, catalog):
    W, b, x, Wh, bh, xh = catalog
    out = []
    for m in range(M):
        s = oracle.score_chain(xh[m], Wh, bh)
        out.append(s)
    return out


_EMB = {}


def _embedding(dev):
    """One (N + 1) x d item-embedding table for every engine of this module (10 GB): N(0, 1) rows, padding row 0."""
    if "emb" not in _EMB:
        g = torch.Generator(device=dev)
        g.manual_seed(77)
        E = torch.randn((N_ITEM + 1, D), generator=g, device=dev, dtype=torch.float32)
        E[0] = 0
        _EMB["emb"] = E
    return _EMB["emb"]


def _scoring_engine(catalog, rank=0, world=1, max_rows=8):
    """Engine over the GPU-resident catalog with a 1-layer dummy decoder: only project.* matters to the scoring
    entry points.  item_embedder.weight must be bound with its full element count: the module's shared table."""
    W, b = catalog[0], catalog[1]
    dev = W.device
    cfg = synth.make_config("tiny", n_item=N_ITEM, emb_dim=D, n_heads=8, n_layers=1, max_len=4, ffn_dim=8, n_user=2)
    small = synth.irn_state_dict(synth.make_config("tiny", n_item=8, emb_dim=D, n_heads=8, n_layers=1, max_len=4,
                                                   ffn_dim=8, n_user=2), seed=1)
    eng = Engine(n_item=N_ITEM, n_user=cfg.n_user, d=D, max_len=cfg.max_len, n_heads=cfg.n_heads, ffn_dim=cfg.ffn_dim,
                 n_layers=1, u_dim=cfg.u_emb_dim, mask_mode=IRS_MASK_IRN, device=dev, max_rows=max_rows, max_seqs=1, max_k=K,
                 rank=rank, world=world)
    sd = {k_: torch.from_numpy(v).to(dev) for k_, v in small.items()
          if k_ not in ("project.weight", "project.bias", "item_embedder.weight")}
    sd["item_embedder.weight"] = _embedding(dev)
    sd["project.weight"], sd["project.bias"] = W, b
    eng.bind_state_dict(sd)  # slices project.* to the shard, packs the bf16 fragments
    return eng


def test_c4_topk_bit_exact_bf16_and_f32(oracle, catalog, oracle_rows):
    W, b, x, Wh, bh, xh = catalog
    eng = _scoring_engine(catalog)
    for sweep in (IRS_SWEEP_BF16, IRS_SWEEP_F32):
        val, ids, st = eng.score_topk(x, K, sweep)
        torch.cuda.synchronize()
        val, ids, st = val.cpu().numpy(), ids.cpu().numpy(), st.cpu().numpy()
        assert (st & 1).sum() == 0, "no row may need the exhaustive fallback on benign data"
        for m in range(M):
            ov, oi = oracle.topk(oracle_rows[m], K)
            assert np.array_equal(ids[m], oi), f"row {m} sweep {sweep}: ids differ"
            assert np.array_equal(val[m].view(np.uint32), ov.view(np.uint32)), f"row {m} sweep {sweep}: values differ"


def test_c4_rank_count_gather_and_lse(oracle, catalog, oracle_rows):
    W, b, x, Wh, bh, xh = catalog
    eng = _scoring_engine(catalog)
    g = np.random.default_rng(4)
    labels = g.integers(0, N_ITEM, size=M).astype(np.int64)
    labels[0] = int(oracle.topk(oracle_rows[0], 30)[1][17])  # a label inside the top-20 window region
    hist = g.integers(0, N_ITEM, size=(M, 100)).astype(np.int64)
    hist[:, :20] = np.stack([oracle.topk(oracle_rows[m], 40)[1][::2] for m in range(M)])  # history items that rank early
    hist[1, 3] = labels[1]
    lt, ht = torch.from_numpy(labels).cuda(), torch.from_numpy(hist).cuda()
    ref = eng.score_gather(x, lt.view(M, 1))[:, 0].contiguous()
    cnt = eng.score_count_before(x, ref, lt, ht).cpu().numpy()
    mx, sm = eng.score_lse(x)
    mx, sm, ref = mx.cpu().numpy(), sm.cpu().numpy(), ref.cpu().numpy()
    for m in range(M):
        s = oracle_rows[m]
        assert ref[m].view(np.uint32) == s[labels[m]].view(np.uint32)
        assert cnt[m] + 1 == oracle.rank_of(s, int(labels[m]), hist[m]), f"row {m}"
        om, osum = oracle.max_sumexp(s)
        assert abs(mx[m] - om) <= 4e-6 * max(1.0, abs(om))  # the LSE sweep sums k in its own order: last-bit differences
        assert abs((mx[m] + np.log(sm[m])) - (om + np.log(osum))) <= 1e-5 * max(1.0, abs(om)), (m, mx[m], sm[m], om, osum)


def test_c4_eight_shards_merge_equals_unsharded(oracle, catalog, oracle_rows):
    """BASELINE config 4's partition: 8 item shards of 1.25M x 256, per-shard top-100, deterministic merge
    (irs_merge_topk) -- equal to the unsharded oracle result bit for bit; rank counts add up across shards."""
    W, b, x, Wh, bh, xh = catalog
    vals, idss, cnts = [], [], []
    g = np.random.default_rng(5)
    labels = torch.from_numpy(g.integers(0, N_ITEM, size=M).astype(np.int64)).cuda()
    ref = None
    for r in range(8):
        eng = _scoring_engine(catalog, rank=r, world=8)
        assert abs(eng.n_local - 1_250_000) <= 32
        v, i, st = eng.score_topk(x, K, IRS_SWEEP_BF16)
        assert not (st & 1).any()
        vals.append(v)
        idss.append(i)
        sc = eng.score_gather(x, labels.view(M, 1))[:, 0]
        ref = sc if ref is None else torch.maximum(ref, sc)
        torch.cuda.synchronize()
        if r < 7:
            del eng
    for r in range(8):  # counts need the assembled label score: second pass, engines rebuilt (one shard resident at a time)
        e2 = _scoring_engine(catalog, rank=r, world=8)
        cnts.append(e2.score_count_before(x, ref.contiguous(), labels, None))
        torch.cuda.synchronize()
        del e2
    mv, mi = eng.merge_topk(torch.stack(vals), torch.stack(idss))
    mv, mi = mv.cpu().numpy(), mi.cpu().numpy()
    cnt = torch.stack(cnts).sum(0).cpu().numpy()
    lab = labels.cpu().numpy()
    for m in range(M):
        ov, oi = oracle.topk(oracle_rows[m], K)
        assert np.array_equal(mi[m], oi) and np.array_equal(mv[m].view(np.uint32), ov.view(np.uint32)), f"row {m}"
        assert cnt[m] + 1 == oracle.rank_of(oracle_rows[m], int(lab[m]), np.zeros(0, dtype=np.int64)), f"row {m}"


M_RING = 512  # >= 256 rows: irs_score_topk takes the compute-bound LDS-DMA ring sweep (k_sweep_ring), the kernel every
              # catalog-scale bench leg and every N = 8 shard runs


def _ring_rows(catalog):
    """512 rows; the first three are rows the oracle has scored over the whole catalog."""
    x = catalog[2]
    g = torch.Generator(device=x.device)
    g.manual_seed(4242)
    xs = torch.randn((M_RING, D), generator=g, device=x.device, dtype=torch.float32)
    xs[:3] = x[:3]
    return xs


def test_c4_ring_kernel_512_rows_single_shard(oracle, catalog, oracle_rows):
    """The ring kernel (PRE + EMIT) at the FULL 10M x 256 catalog with 512 rows -- the 5 GB bf16 fragment image (byte
    offsets past 2^32), several rounds of workgroups, strips 30x longer than any other ring test: every row equal
    to the float32 sweep (another kernel family: k_sweep_f32) bit for bit, 16 sampled rows equal to the
    exhaustive exact kernel, 3 rows equal to the CPU oracle.  Reference: influentialRS.py:418-421."""
    eng = _scoring_engine(catalog, max_rows=M_RING)
    xs = _ring_rows(catalog)
    for rep in range(2):  # the second call runs on a warm workspace (stale candidate counts would show here)
        val, ids, st = eng.score_topk(xs, K, IRS_SWEEP_BF16)
    vf, idf, stf = eng.score_topk(xs, K, IRS_SWEEP_F32)
    torch.cuda.synchronize()
    assert not (st & 1).any() and not (stf & 1).any(), "no row may need the exhaustive fallback on benign data"
    assert torch.equal(ids, idf), "bf16 ring sweep and float32 sweep disagree on ids"
    assert torch.equal(val.view(torch.int32), vf.view(torch.int32)), "bf16 ring sweep and float32 sweep disagree on values"
    sel = torch.tensor([0, 1, 2, 31, 32, 63, 64, 127, 128, 255, 256, 300, 383, 384, 480, 511], device=xs.device)
    ev, ei, _ = eng.score_topk(xs[sel].contiguous(), K, IRS_SWEEP_EXHAUSTIVE)
    torch.cuda.synchronize()
    assert torch.equal(ids[sel], ei) and torch.equal(val[sel].view(torch.int32), ev.view(torch.int32))
    val, ids = val.cpu().numpy(), ids.cpu().numpy()
    for m in range(3):
        ov, oi = oracle.topk(oracle_rows[m], K)
        assert np.array_equal(ids[m], oi) and np.array_equal(val[m].view(np.uint32), ov.view(np.uint32)), f"row {m}"


def test_c4_ring_kernel_512_rows_eight_shards(oracle, catalog, oracle_rows):
    """The same 512 rows through BASELINE config 4's partition: 8 shard engines (1.25M x 256 each, ring kernel),
    packed keys, merge -- equal to the unsharded float32 sweep on every row and to the oracle on its three rows."""
    xs = _ring_rows(catalog)
    full = _scoring_engine(catalog, max_rows=M_RING)
    vf, idf, _ = full.score_topk(xs, K, IRS_SWEEP_F32)
    torch.cuda.synchronize()
    del full
    keys = []
    eng = None
    for r in range(8):
        del eng
        eng = _scoring_engine(catalog, rank=r, world=8, max_rows=M_RING)
        v, i, st = eng.score_topk(xs, K, IRS_SWEEP_BF16)
        assert not (st & 1).any()
        keys.append(eng.pack_topk(v, i))
        torch.cuda.synchronize()
    mv, mi = eng.merge_topk_keys(torch.stack(keys))
    torch.cuda.synchronize()
    assert torch.equal(mi, idf) and torch.equal(mv.view(torch.int32), vf.view(torch.int32))
    mv, mi = mv.cpu().numpy(), mi.cpu().numpy()
    for m in range(3):
        ov, oi = oracle.topk(oracle_rows[m], K)
        assert np.array_equal(mi[m], oi) and np.array_equal(mv[m].view(np.uint32), ov.view(np.uint32)), f"row {m}"


def test_c5_beam32_full_catalog(oracle, catalog):
    """BASELINE config 5 at its FULL shape on one device: beam-width-32 path search of one user over the
    10,000,000 x 256 catalog with the 6-layer L = 200 decoder -- window decode, exact top-100 (bf16 filter),
    exact log-sum-exp over the catalog, beam re-ranking -- 3 steps (the second and third with all 32 beams live)
    against the CPU restatement (oracle_np.beam_search over the same bytes: 65 decodes + 65 chains over 10M items)."""
    W, b, x, Wh, bh, xh = catalog
    dev = W.device
    cfg = synth.make_config("c4")
    assert (cfg.n_item, cfg.emb_dim, cfg.max_len, cfg.n_heads, cfg.n_layers) == (N_ITEM, D, 200, 8, 6)
    small = synth.irn_state_dict(synth.make_config("c4", n_item=8), seed=3)
    host = {k_: v for k_, v in small.items() if k_ not in ("project.weight", "project.bias", "item_embedder.weight")}
    E = _embedding(dev)
    sd = {k_: torch.from_numpy(v).to(dev) for k_, v in host.items()}
    sd["item_embedder.weight"], sd["project.weight"], sd["project.bias"] = E, W, b
    BEAM, P = 32, 3
    eng = Engine(n_item=N_ITEM, n_user=cfg.n_user, d=D, max_len=cfg.max_len, n_heads=cfg.n_heads, ffn_dim=cfg.ffn_dim,
                 n_layers=cfg.n_layers, u_dim=cfg.u_emb_dim, mask_mode=IRS_MASK_IRN, device=dev, max_rows=BEAM,
                 max_seqs=BEAM, max_k=K)
    eng.bind_state_dict(sd)
    g = np.random.default_rng(9)
    L = cfg.max_len
    seqs = np.zeros((1, L), dtype=np.int64)
    seqs[0, L - 1 - 120:] = g.integers(1, N_ITEM + 1, size=121)  # 120 history items + the target slot, pre-padded
    users = np.array([5], dtype=np.int64)
    hep = torch.full((1,), L - 2, dtype=torch.int32, device=dev)
    for graph in (False, True):
        paths, scores, st = eng.beam_search(torch.from_numpy(seqs).to(dev), torch.from_numpy(users).to(dev), hep, P, BEAM,
                                            k=K, sweep=IRS_SWEEP_BF16, use_graph=graph)
        torch.cuda.synchronize()
        if not graph:
            p0, s0 = paths.cpu().numpy(), scores.cpu().numpy()
        else:  # the captured two-step graph and plain stream launches are the same computation
            assert np.array_equal(p0, paths.cpu().numpy()) and np.array_equal(s0, scores.cpu().numpy())
    assert not (st.cpu().numpy() & 1).any()
    host["item_embedder.weight"] = E.cpu().numpy()  # the oracle reads the windows' rows of the same table
    host["project.weight"], host["project.bias"] = Wh, bh
    op, osc = oracle.beam_search(host, cfg, seqs, users, max_path_len=P, gap_len=0, beam=BEAM, k_cand=K)
    assert np.allclose(s0, osc, rtol=0, atol=2e-4), (s0, osc)
    gaps = np.abs(np.diff(osc[0]))
    n_safe = BEAM if gaps.min() > 1e-4 else int(np.argmax(gaps <= 1e-4)) + 1
    assert n_safe >= 8, "the fixture should separate most beams by more than the float32 log-sum-exp noise"
    assert np.array_equal(p0[0, :n_safe], op[0, :n_safe]), (p0[0], op[0])
    assert (np.diff(s0, axis=1) <= 0).all()
