#!/usr/bin/env python3
"""Benchmark of the hot path: one persuasion-path search STEP over a batch of users

    step = decoder over B windows -> row at history end -> score against the whole
           catalog -> top-100 -> window filter + greedy choice + window shift

(IRSNN.get_seq_in_batch's loop body, reference model/influentialRS.py:412-450).
Metric (BASELINE.json): scored user-item pairs/sec = users x n_item / time, whole job.
Default workload = BASELINE.json configs[1]: ml-1m-shaped, d=128, L=200, H=4,
6 layers, F=256, N=3415, synthetic weights and windows, inputs resident in HBM.

    python bench.py                       # 1 GPU, C2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

N > 1 (weak scaling): every rank decodes its own B users, rows are all-gathered
(RCCL), each rank scores all N*B rows against ITS item shard, per-shard top-100
lists are all-gathered and merged identically on every rank.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from influentialrs_amd import synth  # noqa: E402
from influentialrs_amd._lib import (IRS_MASK_IRN, IRS_PROF_ATTN, IRS_PROF_LINEAR, IRS_PROF_NONE,  # noqa: E402
                                    IRS_PROF_REFINE, IRS_PROF_SWEEP, IRS_SWEEP_BF16, IRS_SWEEP_F32)
from influentialrs_amd.engine import Engine  # noqa: E402

PEAK_F32_MATRIX_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 MFMA (exact f32) dense peak
PEAK_BF16_TFLOPS = 2500.0       # dense bf16 MFMA
PEAK_HBM_GBS = 8000.0           # HBM3E spec


def gpu_state_dict(cfg, device, seed=1234):
    """Same init families as synth.irn_state_dict, generated on the device
    (large catalogs: no host round trip)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    d, F, N = cfg.emb_dim, cfg.ffn_dim, cfg.n_item

    def normal(*shape, std=1.0):
        return torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * std

    def uniform(*shape, bound):
        return (torch.rand(*shape, generator=g, device=device, dtype=torch.float32) * 2 - 1) * bound

    sd = {}
    E = normal(N + 1, d)
    E[0] = 0
    sd["item_embedder.weight"] = E
    sd["user_embedder.weight"] = normal(cfg.n_user, cfg.u_emb_dim)
    sd["user_mask_layer.weight"] = uniform(1, cfg.u_emb_dim, bound=cfg.u_emb_dim ** -0.5)
    sd["user_mask_layer.bias"] = normal(1, std=0.1)
    sd["pos_embedder.pe"] = torch.from_numpy(synth.positional_encoding(d, cfg.max_len)).to(device)
    sd["project.weight"] = uniform(N, d, bound=d ** -0.5)
    sd["project.bias"] = normal(N, std=0.1)
    xav = (6.0 / (4 * d)) ** 0.5
    for l in range(cfg.n_layers):
        p = f"decoder.layers.{l}."
        for att in ("self_attn", "multihead_attn"):
            sd[p + att + ".in_proj_weight"] = uniform(3 * d, d, bound=xav)
            sd[p + att + ".in_proj_bias"] = normal(3 * d, std=0.1)
            sd[p + att + ".out_proj.weight"] = uniform(d, d, bound=d ** -0.5)
            sd[p + att + ".out_proj.bias"] = normal(d, std=0.1)
        sd[p + "linear1.weight"] = uniform(F, d, bound=d ** -0.5)
        sd[p + "linear1.bias"] = normal(F, std=0.1)
        sd[p + "linear2.weight"] = uniform(d, F, bound=F ** -0.5)
        sd[p + "linear2.bias"] = normal(d, std=0.1)
        for n in ("norm1", "norm2", "norm3"):
            sd[p + n + ".weight"] = 1.0 + normal(d, std=0.05)
            sd[p + n + ".bias"] = normal(d, std=0.05)
    return sd


def gpu_windows(B, L, n_item, device, seed):
    """ml-1m-shaped evaluation windows: history length log-normal (median 95, sigma 0.95)
    clipped to [18, 2276]; the window keeps the last L-1 history items, pre-padded, target last
    (DataLoaderEvalIRS layout, reference data_provider.py:591-617)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    seqs = torch.randint(1, n_item + 1, (B, L), generator=g, device=device, dtype=torch.int64)
    ln = torch.exp(torch.randn((B, 1), generator=g, device=device) * 0.95 + float(np.log(95.0)))
    hl = ln.clamp(18, 2276).long().clamp(max=L - 1)  # items kept in the window
    col = torch.arange(L, device=device)[None, :]
    seqs[col < (L - 1 - hl)] = 0
    return seqs


def packed_fraction(job):
    """Fraction of the B*L window slots the decoder actually computes (non-pad tokens; the consumed row
    always counts).  The decoder packs them, so executed work scales with this."""
    s = job.seqs
    L = s.shape[1]
    valid = (s != 0)
    valid[torch.arange(s.shape[0], device=s.device), job.hep.long()] = True
    return float(valid.sum().item()) / float(s.numel())


class Job:
    def __init__(self, args, rank, world, device):
        self.args, self.rank, self.world, self.device = args, rank, world, device
        self.cfg = synth.make_config(args.workload)
        if args.n_item:
            self.cfg.n_item = args.n_item
        cfg = self.cfg
        self.B = args.batch
        self.k = 100
        self.sweep = IRS_SWEEP_F32 if args.sweep == "f32" else IRS_SWEEP_BF16
        # Where the path shards: a large catalog is cut into item shards (one exchange step per search
        # step); a catalog that fits every GPU many times over is replicated and the USERS are partitioned, with no
        # data-path collective at all.
        self.sharded = world > 1 and (args.shard == "items" or (args.shard == "auto" and cfg.n_item >= 262144))
        rows = self.B * world if self.sharded else self.B
        self.eng = Engine(n_item=cfg.n_item, n_user=cfg.n_user, d=cfg.emb_dim, max_len=cfg.max_len,
                          n_heads=cfg.n_heads, ffn_dim=cfg.ffn_dim, n_layers=cfg.n_layers, u_dim=cfg.u_emb_dim,
                          mask_mode=IRS_MASK_IRN, device=device, max_rows=rows, max_seqs=self.B, max_k=self.k,
                          rank=rank if self.sharded else 0, world=world if self.sharded else 1)
        if cfg.n_item <= 100_000:
            sd = {k: torch.from_numpy(v).to(device) for k, v in synth.irn_state_dict(cfg, 1234).items()}
        else:
            sd = gpu_state_dict(cfg, device, 1234)
        self.eng.bind_state_dict(sd)
        self.seqs = gpu_windows(self.B, cfg.max_len, cfg.n_item, device, seed=100 + rank)
        self.users = torch.randint(0, cfg.n_user, (self.B,), device=device, dtype=torch.int64)
        self.hep = torch.full((self.B,), cfg.max_len - 2, dtype=torch.int32, device=device)
        self.paths = torch.zeros((self.B, 1), dtype=torch.float32, device=device)
        self.status = torch.zeros(self.B, dtype=torch.int32, device=device)
        if self.sharded:
            self.x_all = torch.empty((rows, cfg.emb_dim), dtype=torch.float32, device=device)
            # exchange of per-shard top-k lists: each rank only needs the lists of ITS rows -> all_to_all of
            # [world, B, k] (score f32 + id i32) instead of all-gathering every row's list on every rank
            self.v_recv = torch.empty((world, self.B, self.k), dtype=torch.float32, device=device)
            self.i_recv = torch.empty((world, self.B, self.k), dtype=torch.int32, device=device)
            self.exchange = args.exchange

    def step(self):
        eng = self.eng
        _, xr, _ = eng.decode(self.seqs, self.users, want_x=False, pos=self.hep)
        if not self.sharded:
            val, ids, _ = eng.score_topk(xr, self.k, self.sweep)
        else:
            import torch.distributed as dist
            dist.all_gather_into_tensor(self.x_all.view(-1), xr.view(-1))
            v, i, _ = eng.score_topk(self.x_all, self.k, self.sweep)  # all rows x this rank's item shard
            i32 = i.to(torch.int32)
            if self.exchange == "all_to_all":
                dist.all_to_all_single(self.v_recv.view(-1), v.view(-1))
                dist.all_to_all_single(self.i_recv.view(-1), i32.view(-1))
            else:  # rehearsal backends without all_to_all: gather everything, keep the own slice
                vg = torch.empty((self.world,) + tuple(v.shape), dtype=v.dtype, device=v.device)
                ig = torch.empty((self.world,) + tuple(i32.shape), dtype=i32.dtype, device=v.device)
                dist.all_gather_into_tensor(vg.view(-1), v.view(-1))
                dist.all_gather_into_tensor(ig.view(-1), i32.view(-1))
                lo = self.rank * self.B
                self.v_recv.copy_(vg[:, lo:lo + self.B])
                self.i_recv.copy_(ig[:, lo:lo + self.B])
            val, ids = eng.merge_topk(self.v_recv, self.i_recv.to(torch.int64))
        eng.path_step(self.seqs, self.hep, val, ids, 0, self.paths, self.status)


def timed(job, steps, world):
    import torch.distributed as dist
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        job.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=job.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def cpu_baseline(cfg, budget_s=20.0):
    """The oracle (CPU restatement, 'port') timed on this box's host cores on a
    bounded sample of the same workload: greedy path steps for a few users, one
    user at a time (the only batch size the published IRN runs at), scoring the
    consumed row only.  Checker code used as a yard-stick, never shipped."""
    import subprocess
    so = os.path.join(REPO, "oracle", "_build", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")], stdout=subprocess.DEVNULL)
    from oracle import oracle_np as O
    sd = synth.irn_state_dict(cfg, 1234)
    seqs = synth.random_windows(4, cfg.max_len, cfg.n_item, seed=3)
    users = np.arange(4)
    hep = cfg.max_len - 2
    W, b = sd["project.weight"], sd["project.bias"]
    done, t0 = 0, time.perf_counter()
    while True:
        for r in range(4):
            x, _ = O.decode(sd, cfg, seqs[r], users[r])
            s = O.score_chain(x[hep], W, b)
            vals, ids0 = O.topk(s, 100)
            nxt = O.select_next(ids0 + 1, vals, seqs[r][:hep + 1])
            seqs[r][:-2] = seqs[r][1:-1].copy()
            seqs[r][-2] = nxt
            done += 1
        if time.perf_counter() - t0 > budget_s or done >= 400:
            break
    dt = time.perf_counter() - t0
    return {"value": done * cfg.n_item / dt, "unit": "pairs/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{done} greedy path steps (4 users, B=1 loop, consumed row only) in {dt:.1f}s: numpy float32 "
                      f"decoder (BLAS threads) + C fma-chain scoring/top-100 (OpenMP)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c3", "c4", "default", "tiny"])
    ap.add_argument("--batch", type=int, default=4096, help="users per rank per step")
    ap.add_argument("--n-item", type=int, default=0, help="override the catalog size")
    ap.add_argument("--sweep", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the N > 1 code path with several ranks on ONE GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--shard", default="auto", choices=["auto", "items", "replicate"],
                    help="N > 1: 'items' = every rank holds a slice of the catalog (rows all-gathered, per-shard top-k "
                         "exchanged, merged); 'replicate' = every rank holds the whole catalog and scores its own users, no "
                         "data-path collective; 'auto' = items from 262144 catalog entries up (a 3415-item catalog is 1.7 MB)")
    ap.add_argument("--exchange", default=None, choices=["all_to_all", "all_gather"],
                    help="how per-shard top-k lists travel (default: all_to_all on nccl, all_gather otherwise)")
    args = ap.parse_args()

    if args.exchange is None:
        args.exchange = "all_to_all" if args.backend == "nccl" else "all_gather"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    device = torch.device("cuda", 0 if args.same_device else local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    job = Job(args, rank, world, device)
    cfg = job.cfg
    for _ in range(args.warmup):
        job.step()
    dt = timed(job, args.steps, world)
    users_total = job.B * world
    value = users_total * cfg.n_item * args.steps / dt

    # second, instrumented pass: HIP events around every launch of each kernel family
    fam = {}
    for name, f in (("linear", IRS_PROF_LINEAR), ("attn", IRS_PROF_ATTN), ("sweep", IRS_PROF_SWEEP), ("refine", IRS_PROF_REFINE)):
        job.eng.prof_enable(f)
        f0 = packed_fraction(job)
        for _ in range(args.steps):
            job.step()
        torch.cuda.synchronize()
        f1 = packed_fraction(job)
        n, ms, fl, by = job.eng.prof_read()
        # decoder kernels run on the packed (non-pad) rows: executed flops = dense-shape flops x packed fraction
        scale = 0.5 * (f0 + f1) if name in ("linear", "attn") else 1.0
        fam[name] = dict(launches=n, ms=ms, flops=fl * scale, bytes=by, packed_fraction=scale)
    job.eng.prof_enable(IRS_PROF_NONE)

    out = None
    if rank == 0:
        dom = max(fam, key=lambda k: fam[k]["ms"])
        f = fam[dom]
        per_launch_ms = f["ms"] / max(f["launches"], 1)
        if dom in ("linear", "attn"):
            ach = f["flops"] / (f["ms"] * 1e-3) / 1e12 if f["ms"] > 0 else 0.0
            roof = {"kernel": {"linear": "k_block + k_linear (decoder fp32 MFMA GEMM family: fused layer kernel, layer-0 QKV)", "attn": "k_attn (decoder attention, fp32 VALU)"}[dom],
                    "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_F32_MATRIX_TFLOPS, "traffic": None}
        else:
            Mrows = users_total
            if dom == "sweep" and Mrows >= 315 and job.sweep == IRS_SWEEP_BF16:
                ach = f["flops"] / (f["ms"] * 1e-3) / 1e12 if f["ms"] > 0 else 0.0
                roof = {"kernel": "k_sweep_bf16 (catalog sweep, bf16 MFMA)", "bound": "mfma", "achieved": ach,
                        "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": None}
            else:
                ach = f["bytes"] / (f["ms"] * 1e-3) / 1e9 if f["ms"] > 0 else 0.0
                roof = {"kernel": f"k_{dom}", "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": ach / PEAK_HBM_GBS, "traffic": None}
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the number is the
        # per-launch mean of the committed rocprofv3 passes of this same command (FETCH_SIZE doubled per the gfx950
        # note + WRITE_SIZE, separate --pmc passes), valid for the default workload only
        pmc_file = os.path.join(REPO, "profiles", "r01", "c2_b4096_pmc_v7.json")
        if dom == "linear" and world == 1 and args.workload == "c2" and args.batch == 4096 and not args.n_item and os.path.exists(pmc_file):
            try:
                with open(pmc_file) as fh:
                    der = json.load(fh)["_derived"]
                roof["traffic"] = float(next(v for k, v in der.items() if "HBM bytes per launch" in k))
                roof["traffic_source"] = ("profiles/r01/c2_b4096_pmc_v7.json: k_block<true,true> (5 of the family's 6 launches per "
                                          "step), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, 2 x FETCH_SIZE + WRITE_SIZE")
                roof["algorithmic_bytes_per_launch"] = 4.0 * 128 * (2 + 1 + 3) * fam[dom]["packed_fraction"] * job.B * cfg.max_len
            except Exception:
                pass
        roof["flops_counted"] = "executed (dense-shape flops x packed non-pad row fraction %.3f)" % fam[dom].get("packed_fraction", 1.0)
        roof["avg_launch_ms"] = per_launch_ms
        roof["launches_per_step"] = f["launches"] / args.steps
        roof["family_ms_per_step"] = {k: v["ms"] / args.steps for k, v in fam.items()}

        lat = lat128 = lat1024 = lat_tokens = None
        if not args.no_latency and world == 1:
            # path-gen p50: one user, 20 greedy steps through irs_generate_paths, stream launches (as the front-end
            # calls it; on ROCm 7.2 the hipGraph replay of the same 180 nodes measures ~3 % slower: 2.82 vs 2.74 ms)
            # the user whose window holds the median number of items of this workload (a window's length sets the
            # decoder's row count, i.e. the latency)
            # fresh windows of the workload's shape (job.seqs has been advanced by every step above: by now those
            # windows are full)
            fresh = gpu_windows(job.B, cfg.max_len, cfg.n_item, device, seed=100 + rank)
            nvalid = (fresh != 0).sum(dim=1)
            iu = int(torch.argsort(nvalid)[nvalid.numel() // 2].item())
            lat_tokens = int(nvalid[iu].item())
            s1 = fresh[iu:iu + 1].clone()
            u1 = job.users[iu:iu + 1].clone()
            h1 = job.hep[iu:iu + 1].clone()
            p1 = torch.zeros((1, 20), dtype=torch.float32, device=device)
            st1 = torch.zeros(1, dtype=torch.int32, device=device)
            def p50(ss, uu, hh, pp, stt, warm, reps):
                ts = []
                ss0, hh0 = ss.clone(), hh.clone()
                for it in range(warm + reps):
                    ss.copy_(ss0)  # every repetition starts from the user's own window (a path search appends its
                    hh.copy_(hh0)  # 20 items to the window it is given: without the reset the windows fill up)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    job.eng.generate_paths(ss, uu, hh, 20, k=100, sweep=job.sweep, use_graph=False, paths=pp, status=stt)
                    torch.cuda.synchronize()
                    if it >= warm:
                        ts.append((time.perf_counter() - t0) * 1e3)
                return float(np.median(ts))

            lat = p50(s1, u1, h1, p1, st1, 20, 200)  # >= 200 repetitions after 20 warm-ups
            def per_user(nb, warm, reps):
                return p50(fresh[:nb].clone(), job.users[:nb].clone(), job.hep[:nb].clone(),
                           torch.zeros((nb, 20), dtype=torch.float32, device=device),
                           torch.zeros(nb, dtype=torch.int32, device=device), warm, reps) / nb

            lat128 = per_user(min(128, job.B), 5, 30)
            lat1024 = per_user(min(1024, job.B), 3, 10)

        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(cfg)

        out = {
            "metric": "scored user-item pairs/sec (whole node)",
            "value": value,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: n_item={cfg.n_item}, d={cfg.emb_dim}, L={cfg.max_len}, H={cfg.n_heads}, "
                                   f"layers={cfg.n_layers}, ffn={cfg.ffn_dim}; one greedy path-search step",
                       "users_per_step": users_total, "users_per_gpu": job.B, "top_k": 100,
                       "windows": "ml-1m-shaped history lengths (log-normal, median 95), pre-padded; decoder skips pad tokens "
                                  "(packed rows), results identical",
                       "packed_row_fraction": fam["linear"]["packed_fraction"],
                       "sweep": args.sweep + (" filter + exact f32 re-score" if args.sweep == "bf16" else ""),
                       "parallelism": "single GPU" if world == 1 else (
                           f"rows data-parallel + item-sharded x{world}: RCCL all-gather of rows, {args.exchange} of per-shard top-100"
                           if job.sharded else
                           f"users partitioned over {world} GPUs, catalog replicated ({cfg.n_item} items): no data-path collective")},
            "path_gen_p50_ms_b1": lat,
            "path_gen_b1_window_tokens": lat_tokens,
            "path_gen_ms_per_user_b128": lat128,
            "path_gen_ms_per_user_b1024": lat1024,
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
