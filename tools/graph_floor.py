#!/usr/bin/env python
"""What a hipGraph would return on the heads region: a chain of N small DEPENDENT kernels launched eagerly on
a stream against the same chain captured once and replayed (torch.cuda.CUDAGraph = hipGraph on ROCm).  The heads region of a step is
~313 launches under 15 us that run back to back at ~6 us each (profiles/r04_step_timeline.txt).
usage: graph_floor.py [N=300]"""
import os
import time

import torch

n = int(os.environ.get("N", 300))
dev = torch.device("cuda")
x = torch.zeros(4096, device=dev)
y = torch.ones(4096, device=dev)


def chain():
    for _ in range(n):
        x.add_(y)                       # each launch depends on the previous one


def wall(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def gpu(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    chain()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    chain()
torch.cuda.synchronize()
for name, fn in (("eager", chain), ("graph replay", g.replay)):
    print("%-13s %d dependent launches: wall %.1f us per launch, stream time %.1f us per launch" % (name, n, wall(fn) / n * 1e6, gpu(fn) / n * 1e6), flush=True)
