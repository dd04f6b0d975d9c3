"""-m gpu: the communicator and the item-sharded search loops below the C ABI (include/irs_hip.h, multi-GPU section)
on the production transport -- RCCL, loaded with dlopen -- with a ONE-rank communicator: what a single-GPU box can
run of that path (library load and symbol resolution, ncclCommInitRank, the collectives on the engine's stream, the
captured step with the collectives inside).  The N > 1 behaviour of the same entry points runs over gloo callbacks in
tests/test_gpu_multirank.py (two ranks on one GPU) and over RCCL wherever two devices exist."""
import numpy as np
import pytest
import torch

from influentialrs_amd import synth
from influentialrs_amd._lib import IRS_SWEEP_BF16
from influentialrs_amd.engine import Comm, IrsError
from gpu_util import make_engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm():
    return Comm(torch.device("cuda:0"), backend="nccl")  # no process group: one rank


def _windows(cfg, n, seed):
    g = np.random.default_rng(seed)
    L = cfg.max_len
    seqs = np.zeros((n, L), dtype=np.int64)
    for i in range(n):
        h = int(g.integers(3, L - 1))
        seqs[i, L - 1 - h:] = g.integers(1, cfg.n_item + 1, size=h + 1)
    return torch.from_numpy(seqs).cuda(), torch.from_numpy(g.integers(0, cfg.n_user, size=n)).cuda()


def test_rccl_one_rank_collectives(comm):
    assert comm.is_rccl and comm.world == 1
    cfg = synth.make_config("tiny")
    eng = make_engine(cfg, synth.irn_state_dict(cfg, 1234), max_rows=16, max_seqs=16)
    x = torch.randn((7, cfg.emb_dim), device="cuda")
    assert torch.equal(eng.allgather_rows(comm, x), x)
    k = torch.randint(0, 2 ** 62, (1, 7, 9), device="cuda", dtype=torch.int64)
    assert torch.equal(eng.exchange_topk(comm, k), k)


@pytest.mark.parametrize("cfgname", ["tiny", "c1"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_sharded_greedy_equals_single_device_loop(comm, cfgname, use_graph):
    """irs_generate_paths_sharded on a one-shard 'world' = irs_generate_paths id for id (the same kernels around two
    RCCL calls per step); with use_graph the step, collectives included, is captured and replayed."""
    cfg = synth.make_config(cfgname)
    eng = make_engine(cfg, synth.irn_state_dict(cfg, 1234), max_rows=16, max_seqs=16)
    seqs, users = _windows(cfg, 9, 5)
    hep = torch.full((9,), cfg.max_len - 2, dtype=torch.int32, device="cuda")
    for rep in range(2):  # the second call reuses the captured graph
        p1, s1 = eng.generate_paths(seqs.clone(), users, hep.clone(), 6, k=100, sweep=IRS_SWEEP_BF16)
        p2, s2 = eng.generate_paths_sharded(comm, seqs.clone(), users, hep.clone(), 6, k=100, sweep=IRS_SWEEP_BF16, use_graph=use_graph)
        torch.cuda.synchronize()
        assert torch.equal(p1, p2) and torch.equal(s1, s2)
    # the step -- RCCL calls included -- really was captured and replayed (bit 0), not silently run as plain launches (bit 1)
    assert eng.lib.irs_sharded_graph_state(eng.h) == (1 if use_graph else 0)


@pytest.mark.parametrize("cfgname,n", [("tiny", 9), ("c1", 2), ("c2", 400)])
def test_sharded_greedy_with_overlapped_collectives_equals_single_device_loop(comm, cfgname, n):
    """irs_set_sharded_overlap (round 5): the step's users as two micro-batches (9 -> 4 + 5), each one's all-gather and all_to_all
    on a side stream chained to the compute stream by events -- here over real RCCL calls (one rank), over gloo callbacks with 3 and
    4 ranks in tests/test_gpu_multirank.py.  Rows are independent: ids and status flags equal irs_generate_paths'.  (c2 x 400
    users: halves of 200 sequences below, the whole batch above the sequence-resident decoder's automatic switch.)"""
    cfg = synth.make_config(cfgname)
    eng = make_engine(cfg, synth.irn_state_dict(cfg, 1234), max_rows=n, max_seqs=n)
    seqs, users = _windows(cfg, n, 5)
    hep = torch.full((n,), cfg.max_len - 2, dtype=torch.int32, device="cuda")
    p1, s1 = eng.generate_paths(seqs.clone(), users, hep.clone(), 5, k=100, sweep=IRS_SWEEP_BF16)
    assert not eng.sharded_overlap
    eng.sharded_overlap = True
    assert eng.sharded_overlap
    for rep in range(2):
        p2, s2 = eng.generate_paths_sharded(comm, seqs.clone(), users, hep.clone(), 5, k=100, sweep=IRS_SWEEP_BF16)
        torch.cuda.synchronize()
        assert torch.equal(p1, p2) and torch.equal(s1, s2)
    eng.sharded_overlap = False


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("use_graph", [False, True])
def test_sharded_beam_equals_single_device_loop(comm, split, use_graph):
    cfg = synth.make_config("tiny")
    eng = make_engine(cfg, synth.irn_state_dict(cfg, 1234), max_rows=32, max_seqs=32)
    seqs, users = _windows(cfg, 3, 8)
    hep = torch.full((3,), cfg.max_len - 2, dtype=torch.int32, device="cuda")
    p1, c1, s1 = eng.beam_search(seqs, users, hep, 5, 4, k=100)
    for rep in range(2):
        p2, c2, s2 = eng.beam_search_sharded(comm, seqs, users, hep, 5, 4, k=100, split_decode=split, use_graph=use_graph)
        torch.cuda.synchronize()
        assert torch.equal(p1, p2) and torch.equal(c1, c2) and torch.equal(s1, s2)


def test_sharded_entry_points_validate(comm):
    cfg = synth.make_config("tiny")
    eng = make_engine(cfg, synth.irn_state_dict(cfg, 1234), max_rows=8, max_seqs=8)
    seqs, users = _windows(cfg, 9, 5)
    hep = torch.full((9,), cfg.max_len - 2, dtype=torch.int32, device="cuda")
    with pytest.raises(IrsError):  # B > max_seqs
        eng.generate_paths_sharded(comm, seqs.clone(), users, hep.clone(), 3)
    shard = make_engine(cfg, synth.irn_state_dict(cfg, 1234), max_rows=32, max_seqs=8, rank=0, world=2)
    with pytest.raises(IrsError, match="communicator is rank"):  # a one-rank communicator on a two-shard context
        shard.generate_paths_sharded(comm, seqs[:4].clone(), users[:4], hep[:4].clone(), 3)


_PROBE = r"""
import sys, faulthandler
faulthandler.dump_traceback_later(100, exit=True)
sys.path.insert(0, {repo!r}); sys.path.insert(0, {repo!r} + "/tests")
import torch
from influentialrs_amd import synth
from influentialrs_amd._lib import IRS_SWEEP_BF16
from influentialrs_amd.engine import Comm
from gpu_util import make_engine
from test_gpu_comm import _windows
dev = torch.device("cuda:0")
comm = Comm(dev, backend="nccl")
print("stage init", comm.lib.irs_comm_rccl_version(), comm.lib.irs_comm_exchange_kind(comm.h), flush=True)
assert comm.lib.irs_comm_rccl_version() // 10000 == 2 and comm.lib.irs_comm_exchange_kind(comm.h) == {kind}
cfg = synth.make_config("tiny")
eng = make_engine(cfg, synth.irn_state_dict(cfg, 1234), max_rows=16, max_seqs=16)
k = torch.randint(0, 2 ** 62, (1, 7, 9), device="cuda", dtype=torch.int64)
assert torch.equal(eng.exchange_topk(comm, k), k)
torch.cuda.synchronize()
print("stage exchange", flush=True)
seqs, users = _windows(cfg, 9, 11)
hep = torch.full((9,), cfg.max_len - 2, dtype=torch.int32, device="cuda")
for use_graph in (False, True):
    p1, s1 = eng.generate_paths(seqs.clone(), users, hep.clone(), 5, k=100, sweep=IRS_SWEEP_BF16)
    p2, s2 = eng.generate_paths_sharded(comm, seqs.clone(), users, hep.clone(), 5, k=100, sweep=IRS_SWEEP_BF16, use_graph=use_graph)
    torch.cuda.synchronize()
    assert torch.equal(p1, p2) and torch.equal(s1, s2)
    print("stage loop graph=%d state=%d" % (use_graph, eng.lib.irs_sharded_graph_state(eng.h)), flush=True)
del eng            # the context (and its captured steps, which hold RCCL nodes) goes before the communicator
torch.cuda.synchronize()
del comm
print("done", flush=True)
"""


@pytest.mark.parametrize("forced", [False, True])
def test_rccl_version_gate_and_forced_send_recv_exchange(forced):
    """The loaded librccl.so speaks the ABI comm.hip was compiled against (rccl.h constants, major version 2, checked at
    load time), and the key exchange's fallback for libraries WITHOUT the ncclAllToAll extension -- grouped ncclSend /
    ncclRecv -- runs when IRS_RCCL_NO_ALLTOALL=1 forces it: same bytes out, and the sharded greedy loop over it equals the
    single-device loop.  Runs in a child process under a time limit: a transport that stalls is a failure of THIS test (with
    the stage it reached), not a hang of the suite."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("IRS_RCCL_NO_ALLTOALL", None)
    if forced:
        env["IRS_RCCL_NO_ALLTOALL"] = "1"
    proc = subprocess.Popen([sys.executable, "-c", _PROBE.format(repo=repo, kind=2 if forced else 1)], env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        out, _ = proc.communicate(timeout=150)
    except subprocess.TimeoutExpired:
        proc.kill()
        out, _ = proc.communicate()
        pytest.fail("exchange probe (forced=%s) did not finish in 150 s; output so far:\n%s" % (forced, out))
    assert proc.returncode == 0 and "done" in out, out


def test_carried_thresholds_inside_the_search_loops_million_items(comm, monkeypatch):
    """irs_generate_paths and irs_generate_paths_sharded on a 1M-item catalog, 10 steps: steps behind the first reuse the previous
    step's emission thresholds (round 5; shards of >= 524288 items, refreshed every 8th step).  An engine created with
    IRS_THR_CARRY=0 (pre-pass + selection on every step) must produce the same paths and status flags id for id, single-device
    loop and sharded loop (one RCCL rank) alike."""
    import bench
    cfg = synth.make_config("c3")
    B, P = 48, 10
    sd = synth.irn_state_dict(cfg, 1234)
    eng = make_engine(cfg, sd, max_rows=B, max_seqs=B)
    monkeypatch.setenv("IRS_THR_CARRY", "0")
    plain = make_engine(cfg, sd, max_rows=B, max_seqs=B)
    monkeypatch.delenv("IRS_THR_CARRY")
    dev = torch.device("cuda:0")
    seqs = bench.gpu_windows(B, cfg.max_len, cfg.n_item, dev, seed=21)
    users = torch.arange(B, device=dev, dtype=torch.int64) * 13 % cfg.n_user
    hep = torch.full((B,), cfg.max_len - 2, dtype=torch.int32, device=dev)
    p0, s0 = plain.generate_paths(seqs.clone(), users, hep.clone(), P, k=100, sweep=IRS_SWEEP_BF16)
    p1, s1 = eng.generate_paths(seqs.clone(), users, hep.clone(), P, k=100, sweep=IRS_SWEEP_BF16)
    p2, s2 = eng.generate_paths_sharded(comm, seqs.clone(), users, hep.clone(), P, k=100, sweep=IRS_SWEEP_BF16)
    torch.cuda.synchronize()
    assert torch.equal(p0, p1) and torch.equal(s0, s1)
    assert torch.equal(p0, p2) and torch.equal(s0, s2)
    assert (p0 > 0).all()  # ten items chosen for every user
