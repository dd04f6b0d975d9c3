"""not-gpu: host-side logic of the drop-in front-end: input contract, checkpoint
key set, shard arithmetic, training-mode (autograd) semantics, error behaviour."""
import numpy as np
import pytest
import torch

from influentialrs_amd import synth
from influentialrs_amd.engine import IrsError, shard_bounds
from influentialrs_amd.model.influentialRS import IRSNN, InfluentialNet
from influentialrs_amd.model.uRS import SampleNet
from influentialrs_amd.model.evaluator import Evaluator
from influentialrs_amd.model.layers import PositionalEncoding, get_end_index, get_item_index, get_start_index


def test_collate_matches_reference_dataloader(golden):
    """synth.collate_eval_irs == DataLoaderEvalIRS._collate_fn (data_provider.py:591-617)."""
    g = golden("contract")
    cfg = synth.make_config("default")
    hists = synth.user_histories(12, cfg.n_item, seed=7)
    rows = synth.eval_rows(hists, cfg.n_item, seed=11)
    for gap in (0, 5):
        raws, seqs, users, targets, labels = synth.collate_eval_irs(rows, cfg.max_len, gap_len=gap)
        assert np.array_equal(seqs, g[f"collate_gap{gap}_seq"])
        assert np.array_equal(users, g[f"collate_gap{gap}_users"])
        assert np.array_equal(targets, g[f"collate_gap{gap}_targets"])
        assert np.array_equal(labels, g[f"collate_gap{gap}_labels"])
        for i, r in enumerate(raws):
            assert np.array_equal(r, g[f"collate_gap{gap}_raw"][i, :g[f"collate_gap{gap}_raw_len"][i]])


def test_state_dict_key_set_is_the_reference_one(golden):
    g = golden("contract")
    net = InfluentialNet(synth.make_config("tiny"))
    sd = net.state_dict()
    assert list(sd.keys()) == list(g["irn_keys"])
    assert [",".join(map(str, v.shape)) for v in sd.values()] == list(g["irn_shapes"])
    snet = SampleNet(synth.make_config("eval_tiny"))
    sd = snet.state_dict()
    assert list(sd.keys()) == list(g["eval_keys"])
    assert [",".join(map(str, v.shape)) for v in sd.values()] == list(g["eval_shapes"])
    # synthetic generator produces exactly that key set, and DataParallel-prefixed checkpoints load
    cfg = synth.make_config("tiny")
    w = synth.irn_state_dict(cfg, 1)
    assert set(w) == set(g["irn_keys"])
    net.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in w.items()})
    assert torch.equal(net.project.bias, torch.from_numpy(w["project.bias"]))


def test_positional_encoding_and_index_helpers():
    pe = PositionalEncoding(16, 12)
    assert np.allclose(pe.pe.numpy(), synth.positional_encoding(16, 12), atol=1e-7)
    assert pe(torch.zeros(3, 5)).shape == (1, 5, 16)
    s = np.array([4, 9, 2, 0, 0])
    assert get_end_index(s) == 2 and get_end_index(np.array([1, 2])) == 1
    assert get_start_index(np.array([0, 0, 5, 6])) == 2
    assert get_item_index(s, 9) == 1 and get_item_index(s, 7) == -1


def test_shard_bounds_partition_the_catalog():
    for n in (1, 31, 32, 33, 3415, 1_000_000, 10_000_000):
        for w in (1, 2, 3, 4, 8):
            if (n + 31) // 32 < w:
                continue
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert all(lo % 32 == 0 for lo, _ in b)
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 32 + 31


def test_training_mode_is_autograd_with_per_row_as_called_mask(oracle):
    """B > 1 in training mode (dropout 0): logits equal the oracle's per-row
    as-called semantics (the published code raises for B > 1)."""
    cfg = synth.make_config("tiny", dropout=0.0)
    sd = synth.irn_state_dict(cfg, 1234)
    net = InfluentialNet(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.train()
    hists = synth.user_histories(8, cfg.n_item, seed=7)
    rows = synth.eval_rows(hists, cfg.n_item, seed=11)[:3]
    raws, seqs, users, targets, labels = synth.collate_eval_irs(rows, cfg.max_len, gap_len=0)
    out = net(torch.from_numpy(seqs), torch.from_numpy(users)).detach().numpy()
    for i in range(3):
        ref = oracle.forward_logits(sd, cfg, seqs[i], users[i], exact=False)
        assert np.abs(out[i] - ref).max() < 2e-5
    irn = IRSNN(cfg, net, "cpu")
    l0 = irn.train_batch(torch.from_numpy(seqs), torch.from_numpy(users))
    l1 = irn.train_batch(torch.from_numpy(seqs), torch.from_numpy(users))
    assert np.isfinite(l0) and l1 < l0


def test_eval_mode_needs_the_hip_engine():
    cfg = synth.make_config("tiny")
    net = InfluentialNet(cfg).eval()
    seqs = torch.zeros((2, cfg.max_len), dtype=torch.int64)
    seqs[:, -1] = 3
    with pytest.raises(IrsError):
        net(seqs, torch.zeros(2, dtype=torch.int64))
    irn = IRSNN(cfg, torch.nn.DataParallel(net), "cpu")  # DataParallel is unwrapped (pipeline.py:43-44)
    assert irn.net is net
    with pytest.raises(IrsError):
        irn.get_seq_in_batch(seqs, torch.zeros(2, dtype=torch.int64), torch.tensor([3, 3]), 2, 0)
    ev = Evaluator(synth.make_config("eval_tiny"), SampleNet(synth.make_config("eval_tiny")).eval(), "cpu")
    with pytest.raises(IrsError):
        ev.get_pp_in_batch(torch.ones((2, 13), dtype=torch.int64), torch.tensor([3, 3]), torch.tensor([2, 2]))


def test_torch_restatement_matches_numpy_oracle(oracle):
    """oracle/oracle_torch.py (the restatement bench.py times as the reference-equivalent CPU baseline) against
    oracle_np.py, which the reference's goldens pin: decoder rows within 2e-5, same greedy step."""
    import torch
    from oracle import oracle_torch as OT
    for name in ("tiny", "c1"):
        cfg = synth.make_config(name)
        sd = synth.irn_state_dict(cfg, 1234)
        irn = OT.TorchIRN(sd, cfg)
        seqs = synth.random_windows(3, cfg.max_len, cfg.n_item, seed=3)
        hep = cfg.max_len - 2
        for r in range(3):
            x, _ = oracle.decode(sd, cfg, seqs[r], r)
            xt = irn.decode(torch.from_numpy(seqs[r]), r).numpy()
            assert np.abs(x - xt).max() < 2e-5
            s = oracle.score_chain(x[hep], sd["project.weight"], sd["project.bias"])
            vals, ids0 = oracle.topk(s, 100)
            nxt = oracle.select_next(ids0 + 1, vals, seqs[r][:hep + 1])
            got, new = irn.path_step_like_reference(torch.from_numpy(seqs[r]), r, hep)
            assert got == nxt and new[-2] == nxt and new[-1] == seqs[r][-1]


def test_cpu_training_trunk_reproduces_the_reference_run(golden):
    """The torch side of training (decoder trunk under autograd, as-called mask with r_u detached, Adam settings)
    against tests/golden/train_tiny.npz = the unmodified reference's own losses.  A module left on the CPU uses the
    reference's nn.Linear + CrossEntropyLoss formulation; the GPU test runs the same golden through the HIP engine."""
    g = golden("train_tiny")
    cfg = synth.make_config("tiny", dropout=0.0)
    net = InfluentialNet(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.irn_state_dict(cfg, 1234).items()})
    irn = IRSNN(cfg, net, "cpu")
    seqs, users = torch.from_numpy(g["irn_seqs"]), torch.from_numpy(g["irn_users"])
    tr = [irn.train_batch(seqs[i:i + 1], users[i:i + 1]) for i in range(4)]
    assert np.allclose(tr, g["irn_train"], rtol=1e-5, atol=0), (tr, g["irn_train"])
    assert np.abs(net.project.bias.detach().numpy() - g["irn_bias_after"]).max() < 1e-4
    assert net.user_embedder.weight.grad is None  # the reference passes r_u through float(): no gradient


def test_bench_windows_are_zipf_without_repeats():
    """bench.py's synthetic windows: pre-padded, target last and absent from the window, history items
    drawn Zipf(1) over the catalog WITHOUT repeats per user, history lengths clipped to the window."""
    import torch
    import bench
    L, N, B = 50, 3415, 256
    s = bench.gpu_windows(B, L, N, torch.device("cpu"), seed=5).numpy()
    assert s.shape == (B, L) and s.min() >= 0 and s.max() <= N
    assert (s[:, -1] > 0).all()
    for r in s:
        v = r[r > 0]
        assert len(set(v.tolist())) == len(v), "an item repeats inside a window"
        nz = np.nonzero(r)[0]
        assert nz[0] == L - len(v), "pads must sit in front (pre-padded window)"
    counts = np.bincount(s[:, :-1].ravel(), minlength=N + 1)[1:]
    assert counts[:10].sum() > 20 * max(counts[1000:1010].sum(), 1) / 10, "head items must dominate (Zipf)"


def test_attention_block_deal_closed_form_equals_the_greedy_longest_first_rule():
    """k_attn16h deals its 16-query blocks (block qb costs qb + 1 key tiles) to four waves longest-first.  The kernel uses the
    closed form of that greedy rule -- the i-th largest block goes to wave i & 7 if that is below 4, else 7 - (i & 7): a
    constant bit pattern reversed into block order (decoder.hip) -- this is the identity it rests on, for every block count."""
    def greedy(nb, nw=4):
        load, mine = [0] * nw, [0] * nw
        for qb in range(nb - 1, -1, -1):
            w = 0
            for v in range(1, nw):
                if load[v] < load[w]:
                    w = v
            load[w] += qb + 1
            mine[w] |= 1 << qb
        return mine

    def closed(nb, wave):
        m = ((0x01010101 << wave) | (0x01010101 << (7 - wave))) & 0xFFFFFFFF
        return int("{:032b}".format(m)[::-1], 2) >> (32 - nb)

    for nb in range(1, 17):
        assert greedy(nb) == [closed(nb, w) for w in range(4)], nb


def test_sequence_resident_plan_half_tile_layout_is_a_mirrored_bijection():
    """k_plan_seq lays a sequence of nb 16-token blocks onto nb consecutive half tiles [slot, slot + nb) of its workgroup by a
    closed form of (slot parity, nb) alone (seq_half_of_block, csrc/decoder.hip; restated here): every block gets exactly one half;
    the two halves of a whole wave tile hold MIRRORED blocks (i, nb - 1 - i), so each such wave gets nb + 1 key tiles of causal
    attention; the one or two halves left over at an odd start / end hold the middle block(s)."""
    def half_of_block(slot_odd, nb, blk):
        mir = nb - 1 - blk
        if not slot_odd:
            if (nb & 1) and blk == mir:
                return nb - 1
            return 2 * blk if blk < mir else 2 * mir + 1
        if nb & 1:
            if blk == mir:
                return 0
            return 1 + 2 * blk if blk < mir else 2 + 2 * mir
        if blk == nb // 2 - 1:
            return 0
        if blk == nb // 2:
            return nb - 1
        return 1 + 2 * blk if blk < mir else 2 + 2 * mir

    for slot in range(0, 16):
        for nb in range(1, 17 - slot):
            halves = [half_of_block(slot & 1, nb, b) for b in range(nb)]
            assert sorted(halves) == list(range(nb)), (slot, nb, halves)
            block_at = {slot + h: b for b, h in enumerate(halves)}
            singles = []
            for t in range(8):
                pair = [block_at.get(2 * t), block_at.get(2 * t + 1)]
                if pair[0] is not None and pair[1] is not None:
                    assert pair[0] + pair[1] == nb - 1, (slot, nb, t, pair)   # mirrored: qb + 1 key tiles each -> nb + 1 per wave
                elif pair[0] is not None or pair[1] is not None:
                    singles.append(pair[0] if pair[0] is not None else pair[1])
            # what is left over at an odd start / end are the middle blocks (about half the attention work of a whole tile each)
            assert len(singles) <= 2 and all(abs(2 * b - (nb - 1)) <= 1 for b in singles), (slot, nb, singles)
