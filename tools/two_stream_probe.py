#!/usr/bin/env python3
"""Experiment (round 4): do two half-batches of a C2 path-search step, issued on two streams, overlap on the GPU?
The fused layer kernel (MFMA + HBM phases) and the attention kernel (instruction-bound, 30 % MFMA) leave different units
idle; a workgroup of one can run beside workgroups of the other on the same CU (LDS 71 + 53 KB, registers 224 + 168).
Prints ms per step: one engine x 4096 users; two engines x 2048 users serially on one stream; the same two on two streams.
usage: python3 tools/two_stream_probe.py [users=4096] [steps=20]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)


def timed(fn, steps=K, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


one = bench.Job("c2", B, 0, 1, dev, "auto", "bf16")
print(f"one engine, {B} users, one stream:          {timed(one.step):.3f} ms per step", flush=True)
del one
halves = [bench.Job("c2", B // 2, r, 1, dev, "auto", "bf16") for r in range(2)]
print(f"two engines x {B // 2}, one stream (serial):   {timed(lambda: [h.step() for h in halves]):.3f} ms per step", flush=True)
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]


def both():
    for h, s in zip(halves, streams):
        with torch.cuda.stream(s):
            h.step()


print(f"two engines x {B // 2}, two streams:            {timed(both):.3f} ms per step", flush=True)
quarters = halves + [bench.Job("c2", B // 2, r + 2, 1, dev, "auto", "bf16") for r in range(2)]
streams += [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]


def four():
    for h, s in zip(quarters, streams):
        with torch.cuda.stream(s):
            h.step()


print(f"four engines x {B // 2} ({2 * B} users), four streams: {timed(four):.3f} ms per step = {timed(four) / 2:.3f} per {B}", flush=True)
