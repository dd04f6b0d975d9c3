"""Time IRSNN.train_batch (one Adam step) with projection + cross entropy on the HIP engine against the reference's
own formulation (nn.Linear + CrossEntropyLoss on materialised logits) on the same module, and report peak memory.
usage: python tools/train_probe.py <cfg> <batch> [n_item]      (GPU box)"""
import copy
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, ".")
from influentialrs_amd import synth  # noqa: E402
from influentialrs_amd.model.influentialRS import IRSNN, InfluentialNet  # noqa: E402
import bench  # noqa: E402

cfgname, B = sys.argv[1], int(sys.argv[2])
over = {"n_item": int(sys.argv[3])} if len(sys.argv) > 3 else {}
cfg = synth.make_config(cfgname, dropout=0.0, **over)
dev = torch.device("cuda:0")
net = InfluentialNet(cfg)
sd = bench.gpu_state_dict(cfg, dev, 1234)
net.to(dev)
net.load_state_dict({k: v for k, v in sd.items()}, strict=False)
seqs = bench.gpu_windows(B, cfg.max_len, cfg.n_item, dev, seed=3)
users = torch.randint(0, cfg.n_user, (B,), device=dev)


def stock_step(net, opt):
    net.train()
    out = net.project(net._decoding_autograd(seqs.clone(), users)[0])[:, :-1, :].contiguous().view(-1, net.n_item)
    tgt = seqs[:, 1:].contiguous().view(-1)
    mask = tgt.gt(0)
    loss = nn.CrossEntropyLoss()(out[mask], tgt[mask] - 1)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss.item()


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(n):
        l = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, torch.cuda.max_memory_allocated() / 2 ** 30, l


irn = IRSNN(cfg, net, dev)
ms, gb, l = timed(lambda: irn.train_batch(seqs, users))
print(f"{cfgname} n_item={cfg.n_item} d={cfg.emb_dim} B={B} rows={B * (cfg.max_len - 1)}: HIP projection+CE  {ms:9.1f} ms/step  peak {gb:6.1f} GiB  loss {l:.4f}", flush=True)
try:
    twin = copy.deepcopy(net)
    opt = torch.optim.Adam(twin.parameters(), betas=(0.9, 0.98), eps=1e-9, lr=cfg.lr1)
    ms2, gb2, l2 = timed(lambda: stock_step(twin, opt))
    print(f"{'':58s}reference formulation {ms2:6.1f} ms/step  peak {gb2:6.1f} GiB  loss {l2:.4f}", flush=True)
except torch.OutOfMemoryError as e:
    print(f"{'':58s}reference formulation: out of memory ({str(e)[:80]})", flush=True)
