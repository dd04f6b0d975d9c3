"""-m gpu: the HIP scoring path (through the C ABI) against the oracle.
Bar: bit-exact ids AND values for every selection (integer/index work and the
fixed-order float chain), 1e-6 relative for log-sum-exp."""
import os

import numpy as np
import pytest
import torch

from influentialrs_amd import synth
from influentialrs_amd._lib import IRS_SWEEP_BF16, IRS_SWEEP_EXHAUSTIVE, IRS_SWEEP_F32
from gpu_util import scoring_only_engine

pytestmark = pytest.mark.gpu


def _weights(n_item, d, seed):
    g = np.random.default_rng(seed)
    W = ((g.random((n_item, d), dtype=np.float32) * 2 - 1) / np.sqrt(d)).astype(np.float32)
    b = (g.standard_normal(n_item) * 0.1).astype(np.float32)
    return W, b


def _rows(M, d, seed):
    g = np.random.default_rng(seed + 100)
    return g.standard_normal((M, d)).astype(np.float32)


CASES = [  # (n_item, d, M, k)
    (257, 16, 6, 100),
    (100, 16, 3, 100),      # exactly k items
    (57, 16, 2, 100),       # fewer than k items
    (3415, 30, 5, 100),     # repo default d (not a multiple of 8)
    (3415, 128, 33, 100),   # ml-1m shaped, ragged row count
    (50_000, 64, 40, 100),
    (200_003, 128, 9, 100),
    (70_001, 256, 5, 32),
]


@pytest.mark.parametrize("n_item,d,M,k", CASES)
@pytest.mark.parametrize("sweep", [IRS_SWEEP_BF16, IRS_SWEEP_F32, IRS_SWEEP_EXHAUSTIVE])
def test_topk_bit_exact(oracle, n_item, d, M, k, sweep):
    W, b = _weights(n_item, d, 1)
    x = _rows(M, d, 2)
    eng = scoring_only_engine(n_item, d, W, b, max_rows=max(M, 1))
    val, ids, st = eng.score_topk(torch.from_numpy(x).cuda(), k, sweep)
    torch.cuda.synchronize()
    val, ids, st = val.cpu().numpy(), ids.cpu().numpy(), st.cpu().numpy()
    for m in range(M):
        s = oracle.score_chain(x[m], W, b)
        ov, oi = oracle.topk(s, k)
        n = len(oi)
        assert np.array_equal(ids[m, :n], oi), f"row {m}: ids differ (sweep {sweep})"
        assert np.array_equal(val[m, :n].view(np.uint32), ov.view(np.uint32)), f"row {m}: values differ"
        if n < k:
            assert (ids[m, n:] == -1).all() and np.isneginf(val[m, n:]).all()
            assert st[m] & 4
    if n_item >= k and sweep != IRS_SWEEP_EXHAUSTIVE:
        assert (st & 1).sum() == 0, "unexpected fallback rows on benign data"


def test_topk_ties_and_fallback(oracle):
    """Massive exact ties (duplicate item rows, constant bias) overflow the
    candidate buffers: the exhaustive kernel must take over and still return
    the (score desc, id asc) order."""
    n_item, d, M, k = 60_000, 32, 4, 100  # 8571 copies of each distinct row: more exact ties than a row's buffers hold
    W, b = _weights(n_item, d, 3)
    W[:] = W[:7][np.arange(n_item) % 7]  # only 7 distinct rows
    b[:] = 0.25
    x = _rows(M, d, 4)
    eng = scoring_only_engine(n_item, d, W, b, max_rows=M)
    for sweep in (IRS_SWEEP_BF16, IRS_SWEEP_F32):
        val, ids, st = eng.score_topk(torch.from_numpy(x).cuda(), k, sweep)
        torch.cuda.synchronize()
        assert (st.cpu().numpy() & 1).all(), "expected the fallback to trigger"
        for m in range(M):
            s = oracle.score_chain(x[m], W, b)
            ov, oi = oracle.topk(s, k)
            assert np.array_equal(ids[m].cpu().numpy(), oi)
            assert np.array_equal(val[m].cpu().numpy().view(np.uint32), ov.view(np.uint32))


def _dup_catalog(n_item, d, n_distinct, seed):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    base = (torch.rand((n_distinct, d), generator=g, device=dev) * 2 - 1) * d ** -0.5
    W = base[torch.arange(n_item, device=dev) % n_distinct].contiguous()
    b = torch.full((n_item,), 0.25, device=dev)
    return W, b, g


@pytest.mark.parametrize("M,k", [(5, 100), (96, 100), (3, 300)])
def test_cooperative_fallback_big_shard_mass_ties(oracle, M, k):
    """A shard of 300,000 items made of 11 distinct rows (27,000 exact copies each): every row overflows its candidate
    lists and is redone exhaustively -- on a shard this size by the COOPERATIVE kernels (k_exh_strips: exact scoring of
    the recorded rows spread over item strips; k_exh_merge: each row's top k out of the strips' lists).  96 rows: more
    than the 64 recorded rows the strips serve, the rest are redone the old way inside k_exh_merge.  Bit-exact against
    the exhaustive yard-stick kernel on every row and against the CPU oracle on three."""
    n_item, d = 300_000, 64
    W, b, g = _dup_catalog(n_item, d, 11, 5)
    x = torch.randn((M, d), generator=g, device=W.device)
    eng = scoring_only_engine(n_item, d, W.cpu().numpy(), b.cpu().numpy(), max_rows=M, max_k=k)
    for sweep in (IRS_SWEEP_BF16, IRS_SWEEP_F32):
        val, ids, st = eng.score_topk(x, k, sweep)
        ev, ei, _ = eng.score_topk(x, k, IRS_SWEEP_EXHAUSTIVE)
        torch.cuda.synchronize()
        assert (st & 1).all(), "expected every row on the exhaustive path"
        assert torch.equal(ids, ei) and torch.equal(val.view(torch.int32), ev.view(torch.int32))
    Wh, bh, xh = W.cpu().numpy(), b.cpu().numpy(), x.cpu().numpy()
    for m in sorted({0, M // 2, M - 1}):
        ov, oi = oracle.topk(oracle.score_chain(xh[m], Wh, bh), k)
        assert np.array_equal(ids[m].cpu().numpy(), oi) and np.array_equal(val[m].cpu().numpy().view(np.uint32), ov.view(np.uint32))


def test_cooperative_fallback_near_duplicate_catalog_time_bound():
    """tools/stress.py's clustered leg as a test: 1,000,000 x 128 items in 120 clusters of ~8300 near-duplicates (2e-3
    apart: inside the bf16 filter's error, more of them than a row's candidate lists hold), 64 rows.  Whatever number of rows lands on the exhaustive path, results
    equal the exhaustive kernel's bit for bit, and a call with flagged rows stays within a time bound that the old
    one-workgroup-per-row fallback (~1.2 ms PER ROW at this size) could not meet for more than a few rows."""
    import time
    n_item, d, M, k = 1_000_000, 128, 64, 100
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    centers = torch.randn((120, d), generator=g, device=dev) * d ** -0.5
    W = centers[torch.randint(0, 120, (n_item,), generator=g, device=dev)] + torch.randn((n_item, d), generator=g, device=dev) * 2e-3
    b = torch.randn(n_item, generator=g, device=dev) * 0.01
    eng = scoring_only_engine(n_item, d, W.cpu().numpy(), b.cpu().numpy(), max_rows=M, max_k=k)
    x = centers[:M] * 6.0 + torch.randn((M, d), generator=g, device=dev) * 0.05  # each row points at one cluster
    val, ids, st = eng.score_topk(x, k, IRS_SWEEP_BF16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        val, ids, st = eng.score_topk(x, k, IRS_SWEEP_BF16)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    ev, ei, _ = eng.score_topk(x, k, IRS_SWEEP_EXHAUSTIVE)
    torch.cuda.synchronize()
    nfb = int((st & 1).sum().item())
    assert torch.equal(ids, ei) and torch.equal(val.view(torch.int32), ev.view(torch.int32))
    assert nfb > 0, "the construction should push rows onto the exhaustive path"
    # measured: 64 flagged rows 5.5 ms = 87 us per row (the one-workgroup-per-row fallback: ~1.2 ms per row, 77 ms)
    assert dt < 0.16e-3 * nfb + 1e-3, f"{nfb} flagged rows took {dt * 1e3:.2f} ms"


def test_topk_thousands_of_ties_without_fallback(oracle):
    """2857 copies of each distinct item row: every row's k-th score is tied thousands of times -- more survivors than
    the 1024 the first refine kept, fewer than the candidate buffers hold.  They are re-scored and ordered in place
    (score desc, id asc); rows that still overflow fall back, all stay exact."""
    n_item, d, M, k = 20_000, 32, 6, 100
    W, b = _weights(n_item, d, 3)
    W[:] = W[:7][np.arange(n_item) % 7]
    b[:] = 0.25
    x = _rows(M, d, 4)
    eng = scoring_only_engine(n_item, d, W, b, max_rows=M)
    val, ids, st = eng.score_topk(torch.from_numpy(x).cuda(), k, IRS_SWEEP_BF16)
    torch.cuda.synchronize()
    assert not (st.cpu().numpy() & 1).all(), "some rows should be handled without the exhaustive fallback"
    for m in range(M):
        ov, oi = oracle.topk(oracle.score_chain(x[m], W, b), k)
        assert np.array_equal(ids[m].cpu().numpy(), oi)
        assert np.array_equal(val[m].cpu().numpy().view(np.uint32), ov.view(np.uint32))


DIRECT_CASES = [  # the small-shard path (n_item <= 4096, k <= 256): one launch up to 32 rows, two up to 1024
    (3415, 128, 1, 100),    # the single-user step
    (3415, 128, 8, 100),
    (4096, 64, 32, 256),    # every limit at once
    (1000, 32, 4, 100),     # sampled threshold with few keys per thread
    (400, 16, 3, 100),      # exactly 4k: the smallest row that takes the sampled threshold
    (399, 16, 3, 100),      # below it: every key kept, bitonic sort of 399
    (130, 8, 2, 1),         # k = 1
    (3415, 30, 5, 100),     # the reference's default d (not a multiple of 4: scalar staging)
    (700, 6, 3, 20),        # d = 2 mod 4
    (3415, 128, 33, 100),   # > 32 rows: scoring (8 rows per workgroup) and selection as two launches
    (3415, 128, 129, 100),  # ragged last row group
    (4096, 64, 256, 256),   # the two-launch form at the key-array and k limits
    (3415, 128, 1000, 100),  # ... and near its row limit (1024)
    (3415, 30, 40, 100),    # scalar staging, two rows per wave
]


@pytest.mark.parametrize("n_item,d,M,k", DIRECT_CASES)
def test_topk_direct_path(oracle, n_item, d, M, k):
    W, b = _weights(n_item, d, 11)
    x = _rows(M, d, 12)
    eng = scoring_only_engine(n_item, d, W, b, max_rows=M, max_k=k)
    for rep in range(2):  # twice: the arrival counters must be left ready for the next call
        val, ids, st = eng.score_topk(torch.from_numpy(x).cuda(), k, IRS_SWEEP_BF16)
        torch.cuda.synchronize()
        val, ids, st = val.cpu().numpy(), ids.cpu().numpy(), st.cpu().numpy()
        assert (st == 0).all()
        for m in range(M):
            ov, oi = oracle.topk(oracle.score_chain(x[m], W, b), k)
            assert np.array_equal(ids[m], oi), f"row {m}: ids differ"
            assert np.array_equal(val[m].view(np.uint32), ov.view(np.uint32)), f"row {m}: values differ"


@pytest.mark.parametrize("n_distinct", [7, 40])
def test_topk_direct_path_ties(oracle, n_distinct):
    """Duplicate item rows on the small-shard path: with 7 distinct rows the ties at the threshold overflow the
    survivor buffer (full sort of the row's keys), with 40 they do not; the order stays (score desc, id asc)."""
    n_item, d, M, k = 3000, 32, 4, 100
    W, b = _weights(n_item, d, 3)
    W[:] = W[:n_distinct][np.arange(n_item) % n_distinct]
    b[:] = 0.25
    x = _rows(M, d, 4)
    eng = scoring_only_engine(n_item, d, W, b, max_rows=M)
    val, ids, st = eng.score_topk(torch.from_numpy(x).cuda(), k, IRS_SWEEP_BF16)
    torch.cuda.synchronize()
    for m in range(M):
        ov, oi = oracle.topk(oracle.score_chain(x[m], W, b), k)
        assert np.array_equal(ids[m].cpu().numpy(), oi)
        assert np.array_equal(val[m].cpu().numpy().view(np.uint32), ov.view(np.uint32))


@pytest.mark.parametrize("n_item,d,M", [(257, 16, 5), (3415, 30, 7), (3415, 128, 40), (40_000, 256, 3)])
def test_dense_and_gather_bitwise(oracle, n_item, d, M):
    """fp32 MFMA sweep == VALU chain == oracle chain, bit for bit."""
    W, b = _weights(n_item, d, 5)
    x = _rows(M, d, 6)
    eng = scoring_only_engine(n_item, d, W, b, max_rows=M)
    xt = torch.from_numpy(x).cuda()
    dense = eng.score_dense(xt).cpu().numpy()
    g = np.random.default_rng(0)
    ids = g.integers(0, n_item, size=(M, 17)).astype(np.int64)
    ids[0, 0] = -1
    got = eng.score_gather(xt, torch.from_numpy(ids).cuda()).cpu().numpy()
    for m in range(M):
        s = oracle.score_chain(x[m], W, b)
        assert np.array_equal(dense[m].view(np.uint32), s.view(np.uint32)), f"dense row {m}"
        for j in range(17):
            if ids[m, j] < 0:
                assert np.isneginf(got[m, j])
            else:
                assert got[m, j].view(np.uint32) == s[ids[m, j]].view(np.uint32)


@pytest.mark.parametrize("n_item,d,M", [(257, 16, 6), (3415, 128, 10), (60_000, 64, 4)])
def test_count_before_matches_rank(oracle, n_item, d, M):
    W, b = _weights(n_item, d, 7)
    x = _rows(M, d, 8)
    eng = scoring_only_engine(n_item, d, W, b, max_rows=M)
    xt = torch.from_numpy(x).cuda()
    g = np.random.default_rng(1)
    labels = g.integers(0, n_item, size=M).astype(np.int64)
    hist = g.integers(0, n_item, size=(M, 23)).astype(np.int64)
    hist[:, 3] = hist[:, 4]  # duplicates
    hist[:, 5] = -1          # unused slot
    hist[1, 6] = labels[1]   # the label itself is never excluded by the count
    ref = eng.score_gather(xt, torch.from_numpy(labels[:, None]).cuda())[:, 0].contiguous()
    cnt = eng.score_count_before(xt, ref, torch.from_numpy(labels).cuda(), torch.from_numpy(hist).cuda()).cpu().numpy()
    for m in range(M):
        s = oracle.score_chain(x[m], W, b)
        h = hist[m][hist[m] >= 0]
        assert cnt[m] + 1 == oracle.rank_of(s, int(labels[m]), h), f"row {m}"


@pytest.mark.parametrize("n_item,d,M", [(257, 16, 6), (3415, 128, 10), (100_000, 64, 3)])
def test_lse(oracle, n_item, d, M):
    W, b = _weights(n_item, d, 9)
    x = _rows(M, d, 10)
    eng = scoring_only_engine(n_item, d, W, b, max_rows=M)
    mx, sm = eng.score_lse(torch.from_numpy(x).cuda())
    mx, sm = mx.cpu().numpy(), sm.cpu().numpy()
    for m in range(M):
        s = oracle.score_chain(x[m], W, b)
        om, osum = oracle.max_sumexp(s)
        assert abs(mx[m] - om) <= 4e-6 * max(1.0, abs(om))  # the LSE sweep sums k in its own order: last-bit differences
        # (max, sum) is one number, max + log(sum): a max that differs in the last bit rescales the sum
        assert abs((mx[m] + np.log(sm[m])) - (om + np.log(osum))) <= 4e-6 * max(1.0, abs(om))


@pytest.mark.parametrize("n_item,d,M", [(3415, 128, 10), (200_000, 64, 3), (300_017, 128, 70), (150_000, 256, 33), (120_000, 40, 5)])
def test_topk_lse_fused_equals_separate_calls(oracle, n_item, d, M):
    """irs_score_topk_lse (one pass over the float32 catalog emits candidates AND accumulates max / sum exp on the
    swept path) == irs_score_topk + irs_score_lse: ids and values bit for bit, the log-sum-exp to float32 rounding;
    row 0 also against the oracle."""
    W, b = _weights(n_item, d, 21)
    x = _rows(M, d, 22)
    eng = scoring_only_engine(n_item, d, W, b, max_rows=M)
    xt = torch.from_numpy(x).cuda()
    for sweep in (IRS_SWEEP_BF16, IRS_SWEEP_F32):
        v0, i0, s0 = eng.score_topk(xt, 100, sweep)
        m0, e0 = eng.score_lse(xt)
        v1, i1, s1, m1, e1 = eng.score_topk_lse(xt, 100, sweep)
        torch.cuda.synchronize()
        assert torch.equal(i0, i1) and torch.equal(v0.view(torch.int32), v1.view(torch.int32))
        assert not (s1.cpu().numpy() & 1).any()
        l0 = (m0.double() + e0.double().log()).cpu().numpy()
        l1 = (m1.double() + e1.double().log()).cpu().numpy()
        assert np.abs(l0 - l1).max() <= 4e-6 * max(1.0, np.abs(l0).max())
    s = oracle.score_chain(x[0], W, b)
    ov, oi = oracle.topk(s, 100)
    assert np.array_equal(i1[0].cpu().numpy(), oi) and np.array_equal(v1[0].cpu().numpy().view(np.uint32), ov.view(np.uint32))
    om, osum = oracle.max_sumexp(s)
    assert abs(l1[0] - (om + np.log(osum))) <= 4e-6 * max(1.0, abs(om))


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_merge_equals_unsharded(oracle, world):
    """Item-sharded top-k + merge == single-shard top-k, bit for bit."""
    n_item, d, M, k = 30_011, 64, 12, 100
    W, b = _weights(n_item, d, 11)
    x = _rows(M, d, 12)
    xt = torch.from_numpy(x).cuda()
    vals, idss = [], []
    for r in range(world):
        eng = scoring_only_engine(n_item, d, W, b, max_rows=M, rank=r, world=world)
        v, i, _ = eng.score_topk(xt, k, IRS_SWEEP_BF16)
        vals.append(v)
        idss.append(i)
    mv, mi = eng.merge_topk(torch.stack(vals), torch.stack(idss))
    mv, mi = mv.cpu().numpy(), mi.cpu().numpy()
    for m in range(M):
        s = oracle.score_chain(x[m], W, b)
        ov, oi = oracle.topk(s, k)
        assert np.array_equal(mi[m], oi)
        assert np.array_equal(mv[m].view(np.uint32), ov.view(np.uint32))


def test_bf16_filter_survives_worst_case_rounding(oracle):
    """Adversarial inputs for the bf16 filter's error bound (DESIGN.md "Why the bf16 filter is exact"): every entry
    of x and of the 'hot' item rows sits exactly on a bf16 rounding midpoint, with the signs of the rounding errors
    aligned -- coordinates 0..63 of x round DOWN, 64..127 round UP; 'down' items live on the first half (their
    weights round down too), 'up' items on the second half (weights round up).  The bf16 scores of the up items are
    then 2^-7 relative ABOVE their exact scores and those of the down items 2^-7 BELOW, the full 2u of the worst
    case.  Biases make the down items the TRUE top-k by a small margin while every one of them scores below every up
    item in bf16 -- by more than the (half-width) bound the first version used, which silently returned up items.
    The up items sit in even tiles (the tiles the sampled pre-pass sees), so the emission threshold is set by them."""
    n_item, d, M, k = 70_001, 128, 3, 100
    g = np.random.default_rng(21)
    W = ((g.random((n_item, d), dtype=np.float32) * 2 - 1) * 0.05).astype(np.float32)
    b = (g.standard_normal(n_item) * 0.1).astype(np.float32)
    lo, hi = np.float32(1 + 2.0 ** -8), np.float32(1 + 3 * 2.0 ** -8)  # midpoints: RNE -> 1.0 (down), 1 + 2^-6 (up)
    assert oracle.bf16_round(np.array([lo, hi]))[0] == 1.0 and oracle.bf16_round(np.array([lo, hi]))[1] == np.float32(1 + 2.0 ** -6)
    x = np.zeros((M, d), dtype=np.float32)
    x[:, :64], x[:, 64:] = lo, hi
    tiles = g.permutation(n_item // 32 - 1)
    up_tiles = [t for t in tiles if t % 2 == 0][:300]
    dn_tiles = [t for t in tiles if t % 2 == 1][:300]
    up = np.array([t * 32 + int(g.integers(0, 32)) for t in up_tiles])
    dn = np.array([t * 32 + int(g.integers(0, 32)) for t in dn_tiles])
    W[up] = 0
    W[up, 64:] = hi
    W[dn] = 0
    W[dn, :64] = lo
    base_up, base_dn = 64.0 * float(hi) * float(hi), 64.0 * float(lo) * float(lo)
    b[up] = 0.0
    b[dn] = np.float32(base_up - base_dn + 0.13) + (np.arange(300) * 1e-4).astype(np.float32)
    eng = scoring_only_engine(n_item, d, W, b, max_rows=M)
    xt = torch.from_numpy(x).cuda()
    val, ids, st = eng.score_topk(xt, k, IRS_SWEEP_BF16)
    ev, ei, _ = eng.score_topk(xt, k, IRS_SWEEP_EXHAUSTIVE)
    torch.cuda.synchronize()
    val, ids, st = val.cpu().numpy(), ids.cpu().numpy(), st.cpu().numpy()
    s = oracle.score_chain(x[0], W, b)
    ov, oi = oracle.topk(s, k)
    assert set(oi.tolist()) <= set(dn.tolist()), "construction: the true top-k are 'down' items"
    a_up = 64.0 * (1 + 2.0 ** -6) ** 2
    a_dn_max = 64.0 + float(b[dn].max())
    assert a_dn_max < a_up - 0.8, "construction: in bf16 every down item scores well below every up item"
    for m in range(M):
        assert np.array_equal(ids[m], oi) and np.array_equal(val[m].view(np.uint32), ov.view(np.uint32)), f"row {m}"
        assert np.array_equal(ei[m].cpu().numpy(), oi)
    assert (st & 1).sum() == 0, "handled by the filter itself, not by the exhaustive fallback"


RING_CASES = [  # >= 256 rows: the compute-bound LDS-DMA ring sweep (PRE and EMIT), both row-block sizes, ragged row counts
    (50_000, 128, 300, 100),    # 10 row tiles: 512-row blocks waste more than 256-row ones -> RT = 2, two tiles per step
    (50_000, 128, 513, 100),    # 17 row tiles: one dead row tile short of two 512-row blocks... ragged both ways
    (70_001, 256, 260, 50),     # d_pad = 256: two row tiles per wave, one tile per step
    (20_000, 64, 1024, 100),    # d_pad = 64
    (9_000, 30, 256, 100),      # the reference's default d (padded to 32), exactly one 256-row block
    (40_000, 16, 1000, 7),      # a single k-step per tile: the ring advances behind the step, not inside it
    (300_000, 128, 1024, 100),  # long strips, several rounds of workgroups
]


def test_ring_kernel_c3_bench_shape(oracle):
    """BASELINE configs[2]'s scoring shape exactly as bench.py's C3 leg and `scoring` object run it: 1,000,000 x 128,
    1024 rows through the ring kernel -- every row equal to the float32 sweep bit for bit, 16 sampled rows equal to the
    exhaustive exact kernel, 3 rows equal to the CPU oracle's chain + top-k."""
    n_item, d, M, k = 1_000_000, 128, 1024, 100
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    Wt = (torch.rand((n_item, d), generator=g, device=dev, dtype=torch.float32) * 2 - 1) * d ** -0.5
    bt = torch.randn((n_item,), generator=g, device=dev, dtype=torch.float32) * 0.1
    xt = torch.randn((M, d), generator=g, device=dev, dtype=torch.float32)
    W, b = Wt.cpu().numpy(), bt.cpu().numpy()
    eng = scoring_only_engine(n_item, d, W, b, max_rows=M, max_k=k)
    for rep in range(2):
        val, ids, st = eng.score_topk(xt, k, IRS_SWEEP_BF16)
    vf, idf, stf = eng.score_topk(xt, k, IRS_SWEEP_F32)
    sel = torch.arange(0, M, 64, device=dev)
    ev, ei, _ = eng.score_topk(xt[sel].contiguous(), k, IRS_SWEEP_EXHAUSTIVE)
    torch.cuda.synchronize()
    assert not (st & 1).any() and not (stf & 1).any()
    assert torch.equal(ids, idf) and torch.equal(val.view(torch.int32), vf.view(torch.int32))
    assert torch.equal(ids[sel], ei) and torch.equal(val[sel].view(torch.int32), ev.view(torch.int32))
    x = xt.cpu().numpy()
    val, ids = val.cpu().numpy(), ids.cpu().numpy()
    for m in (0, 511, 1023):
        ov, oi = oracle.topk(oracle.score_chain(x[m], W, b), k)
        assert np.array_equal(ids[m], oi) and np.array_equal(val[m].view(np.uint32), ov.view(np.uint32)), f"row {m}"


@pytest.mark.parametrize("n_item,d,M,k", RING_CASES)
def test_topk_many_rows_ring_kernel(oracle, n_item, d, M, k):
    """The ring kernel against the exhaustive exact kernel (every row, bit for bit) and the CPU oracle (a few rows)."""
    W, b = _weights(n_item, d, 31)
    x = _rows(M, d, 32)
    eng = scoring_only_engine(n_item, d, W, b, max_rows=M, max_k=k)
    xt = torch.from_numpy(x).cuda()
    for rep in range(2):
        val, ids, st = eng.score_topk(xt, k, IRS_SWEEP_BF16)
    ev, ei, _ = eng.score_topk(xt, k, IRS_SWEEP_EXHAUSTIVE)
    torch.cuda.synchronize()
    assert torch.equal(ids, ei) and torch.equal(val.view(torch.int32), ev.view(torch.int32))
    assert (st.cpu().numpy() & 1).sum() == 0, "unexpected fallback rows on benign data"
    val, ids = val.cpu().numpy(), ids.cpu().numpy()
    for m in (0, M // 2, M - 1):
        ov, oi = oracle.topk(oracle.score_chain(x[m], W, b), k)
        assert np.array_equal(ids[m], oi) and np.array_equal(val[m].view(np.uint32), ov.view(np.uint32)), f"row {m}"


def test_topk_many_rows_ties_and_dense_hits(oracle):
    """The ring kernel's dense-hit path (more than 8 lanes of a tile above the threshold: duplicate item rows make whole
    tiles tie) and the overflow fallback behind it, at 256+ rows."""
    n_item, d, M, k = 40_000, 32, 288, 100
    W, b = _weights(n_item, d, 3)
    W[:] = W[:7][np.arange(n_item) % 7]  # only 7 distinct rows
    b[:] = 0.25
    x = _rows(M, d, 4)
    x[5] = x[4]  # identical rows too
    eng = scoring_only_engine(n_item, d, W, b, max_rows=M)
    xt = torch.from_numpy(x).cuda()
    val, ids, st = eng.score_topk(xt, k, IRS_SWEEP_BF16)
    ev, ei, _ = eng.score_topk(xt, k, IRS_SWEEP_EXHAUSTIVE)
    torch.cuda.synchronize()
    assert (st.cpu().numpy() & 1).all(), "expected the fallback to trigger"
    assert torch.equal(ids, ei) and torch.equal(val.view(torch.int32), ev.view(torch.int32))
    ov, oi = oracle.topk(oracle.score_chain(x[4], W, b), k)
    assert np.array_equal(ids[4].cpu().numpy(), oi) and np.array_equal(ids[5].cpu().numpy(), oi)


def _random_shapes(n, seed):
    g = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        n_item = int(np.exp(g.uniform(np.log(40), np.log(400_000))))
        d = int(g.choice([8, 16, 24, 30, 32, 40, 64, 96, 128, 160, 192, 256]))
        M = int(np.exp(g.uniform(0, np.log(600))))
        k = int(g.integers(1, 101))
        out.append((n_item, d, M, k))
    return out


@pytest.mark.parametrize("n_item,d,M,k", _random_shapes(int(os.environ.get("IRS_RANDOM_SHAPES", "28")),
                                                        int(os.environ.get("IRS_RANDOM_SHAPES_SEED", "20261004"))))
def test_topk_random_shapes_against_exhaustive(n_item, d, M, k):
    """Randomly drawn (catalog, width, rows, k): launch geometry, ragged tiles, strip counts, direct / streaming / ring
    paths as they fall -- the filtered top-k (bf16 and fused with the log-sum-exp) equals the exhaustive exact kernel
    bit for bit on every row; scores are heavy-tailed (a tenth of the rows scaled by 8, a few items with a large bias)."""
    W, b = _weights(n_item, d, n_item + d)
    g = np.random.default_rng(M + k)
    b[g.integers(0, n_item, size=max(1, n_item // 5000))] += 3.0
    x = _rows(M, d, M + 7 * k)
    x[g.random(M) < 0.1] *= 8.0
    eng = scoring_only_engine(n_item, d, W, b, max_rows=M, max_k=k)
    xt = torch.from_numpy(x).cuda()
    ev, ei, es = eng.score_topk(xt, k, IRS_SWEEP_EXHAUSTIVE)
    val, ids, st = eng.score_topk(xt, k, IRS_SWEEP_BF16)
    v2, i2, s2, mx, sm = eng.score_topk_lse(xt, k, IRS_SWEEP_BF16)
    m0, e0 = eng.score_lse(xt)
    torch.cuda.synchronize()
    assert torch.equal(ids, ei) and torch.equal(val.view(torch.int32), ev.view(torch.int32))
    assert torch.equal(i2, ei) and torch.equal(v2.view(torch.int32), ev.view(torch.int32))
    assert torch.equal(st & 4, es & 4) and torch.equal(s2 & 4, es & 4)  # "fewer than k items" agrees
    l0 = (m0.double() + e0.double().log()).cpu().numpy()
    l1 = (mx.double() + sm.double().log()).cpu().numpy()
    assert np.abs(l0 - l1).max() <= 4e-6 * max(1.0, np.abs(l0).max())


@pytest.mark.parametrize("n_item,d,M,k", _random_shapes(int(os.environ.get("IRS_RANDOM_SHAPES", "16")),
                                                        int(os.environ.get("IRS_RANDOM_SHAPES_SEED", "7"))))
def test_rank_gather_dense_random_shapes_agree(n_item, d, M, k):
    """The other scoring entry points on randomly drawn shapes, against each other (all of them evaluate the same
    fixed-order float32 chain): gather == the dense logits at those ids, bit for bit; count_before == the rank
    counted on the dense logits (score desc, id asc, excluded ids skipped); the top-k's values == dense at its ids;
    max + log(sum exp) == the same over the dense logits."""
    M = min(M, 48)
    W, b = _weights(n_item, d, n_item + 3 * d)
    x = _rows(M, d, M + 11 * k)
    eng = scoring_only_engine(n_item, d, W, b, max_rows=M, max_k=k)
    xt = torch.from_numpy(x).cuda()
    dense = eng.score_dense(xt)                                   # [M, n_item]
    g = torch.Generator(device="cuda")
    g.manual_seed(n_item + M)
    ids = torch.randint(0, n_item, (M, 7), generator=g, device="cuda")
    got = eng.score_gather(xt, ids)
    assert torch.equal(got.view(torch.int32), dense.gather(1, ids).view(torch.int32))
    ref_id = ids[:, 0].contiguous()
    ref = got[:, 0].contiguous()
    excl = torch.randint(0, n_item, (M, 5), generator=g, device="cuda")
    excl[:, 0] = torch.argmax(dense, dim=1)                       # an item that certainly ranks before most references
    cnt = eng.score_count_before(xt, ref, ref_id, excl)
    ar = torch.arange(n_item, device="cuda")[None, :]
    before = (dense > ref[:, None]) | ((dense == ref[:, None]) & (ar < ref_id[:, None]))
    mask = torch.zeros_like(before)
    mask.scatter_(1, excl, True)
    want = (before & ~mask).sum(1)
    assert torch.equal(cnt, want)
    val, tid, st = eng.score_topk(xt, k, IRS_SWEEP_BF16)
    live = tid >= 0
    assert torch.equal(val[live].view(torch.int32), dense.gather(1, tid.clamp(min=0))[live].view(torch.int32))
    mx, sm = eng.score_lse(xt)
    want_lse = torch.logsumexp(dense.double(), dim=1)
    assert ((mx.double() + sm.double().log()) - want_lse).abs().max().item() <= 4e-6 * max(1.0, want_lse.abs().max().item())


@pytest.mark.parametrize("n_item,d,M", [(300_017, 128, 32), (150_001, 256, 32), (150_001, 256, 9), (40_000, 128, 1)])
def test_lse_ring_equals_register_form(oracle, monkeypatch, n_item, d, M):
    """At most 32 rows at d = 128 / 256 (a beam search's rows) the log-sum-exp pass over the float32 catalog -- alone
    (irs_score_lse) and fused with the emission (irs_score_topk_lse) -- runs through per-wave LDS-DMA rings (k_lse_ring).
    IRS_LSE_RING=0 at context creation keeps the register-fragment kernel (k_lse_f32) for those shapes too.  Both sum a
    score in the same order, so the top-k (ids, value bits, status) must be identical and the (max, sum) pairs equal to
    the partial sums' regrouping (the ring form runs 2048 waves, the register form 1024: different partials); one row
    also against the oracle."""
    W, b = _weights(n_item, d, 31)
    x = _rows(M, d, 32)
    xt = torch.from_numpy(x).cuda()
    res = {}
    for ring in ("1", "0"):
        monkeypatch.setenv("IRS_LSE_RING", ring)
        eng = scoring_only_engine(n_item, d, W, b, max_rows=32)
        monkeypatch.delenv("IRS_LSE_RING")
        v, i, s, m, e = eng.score_topk_lse(xt, 100, IRS_SWEEP_BF16)
        m2, e2 = eng.score_lse(xt)
        torch.cuda.synchronize()
        res[ring] = (v.clone(), i.clone(), s.clone(), (m.double() + e.double().log()).cpu().numpy(), (m2.double() + e2.double().log()).cpu().numpy(),
                     m.clone())
        del eng
    a, c = res["1"], res["0"]
    assert torch.equal(a[1], c[1]) and torch.equal(a[0].view(torch.int32), c[0].view(torch.int32)) and torch.equal(a[2], c[2])
    assert torch.equal(a[5], c[5]), "the rows' maxima are exact in both forms"
    assert np.abs(a[3] - c[3]).max() <= 2e-6 * max(1.0, np.abs(c[3]).max()) and np.abs(a[4] - c[4]).max() <= 2e-6 * max(1.0, np.abs(c[4]).max())
    s0 = oracle.score_chain(x[0], W, b)
    ov, oi = oracle.topk(s0, 100)
    assert np.array_equal(a[1][0].cpu().numpy(), oi) and np.array_equal(a[0][0].cpu().numpy().view(np.uint32), ov.view(np.uint32))
    om, osum = oracle.max_sumexp(s0)
    assert abs(a[3][0] - (om + np.log(osum))) <= 4e-6 * max(1.0, abs(om))


@pytest.mark.parametrize("n_item,d,M", [(600_017, 128, 70), (1_000_000, 128, 600), (150_000, 256, 33), (60_000, 64, 300)])
def test_carried_emission_thresholds_stay_exact(n_item, d, M):
    """irs_score_topk_carry (round 5): the pre-pass and the threshold selection are skipped and the previous call's emission
    thresholds reused.  (a) rows that ARE the previous rows a little later (a small perturbation, as one path-search step is):
    exact lists, no row on the exhaustive path; (b) rows that have NOTHING to do with the previous call's (fresh random rows, rows
    scaled by 1/4 and by 4: thresholds far too high and far too low): still the exhaustive kernel's lists bit for bit -- the
    validation in k_refine sends the rows whose threshold no longer fits to the exhaustive path (IRS_ROW_FALLBACK); (c) a plain
    call afterwards selects fresh thresholds again (no row falls back).  Shards below 524288 items (the last two cases) never carry:
    their pre-pass sees every tile and selects the tight k-th group maximum, which the next step's rows would often miss."""
    W, b = _weights(n_item, d, n_item + d)
    eng = scoring_only_engine(n_item, d, W, b, max_rows=M, max_k=100)
    g = np.random.default_rng(M + d)
    x0 = _rows(M, d, 5)

    def check(x, carry):
        xt = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        ev, ei, _ = eng.score_topk(xt, 100, IRS_SWEEP_EXHAUSTIVE)
        val, ids, st = eng.score_topk(xt, 100, IRS_SWEEP_BF16, carry=carry)
        torch.cuda.synchronize()
        assert torch.equal(ids, ei) and torch.equal(val.view(torch.int32), ev.view(torch.int32))
        return int((st & 1).sum().item())

    assert check(x0, False) == 0
    x = x0
    for step in range(10):  # (the default period refreshes the thresholds on the 8th call)
        x = (x + 0.02 * g.standard_normal(x.shape)).astype(np.float32)
        assert check(x, True) == 0, step
    fb = check(_rows(M, d, 99), True)            # unrelated rows: exact, some on the exhaustive path
    fb_low = check(0.25 * _rows(M, d, 98), True)  # every threshold far too high
    fb_high = check(4.0 * _rows(M, d, 97), True)  # every threshold far too low (buckets overflow)
    assert (fb_low > 0 or fb_high > 0 or fb > 0) == (n_item >= 524288), (fb, fb_low, fb_high)
    assert check(_rows(M, d, 96), False) == 0
