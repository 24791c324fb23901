#!/usr/bin/env python3
"""Probe: does running the batch as K concurrent micro-batches (K caller streams, one process, shared side lanes) raise the rate of the
airfoil B=8 training step?  Timing only -- the halves are independent steps here (no shared loss / gradient sum).
   python profiles/microbatch_probe.py [f32|bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import bsms_gnn_amd as eng

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"

def make(B):
    wl = bench.build_workload("airfoil", B, "cuda")
    torch.manual_seed(0)
    sim = eng.BSMS_Simulator(bench.make_cfg(wl["cfg"])).cuda()
    data = bench.data_tuple(wl)
    sim(data, True, True)
    sim.process.precision = prec
    return eng.DataParallel(sim), data

def rate(parts, n=60):
    """parts: [(dp, data, stream)]; one 'step' = every part once."""
    def once():
        for dp, data, st in parts:
            with torch.cuda.stream(st):
                dp.step_loss_backward(data, True)
    for _ in range(10):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

cur = torch.cuda.current_stream()
dp8, d8 = make(8)
print(f"{prec}: B=8 one stream            {rate([(dp8, d8, cur)]):7.3f} ms per 8 samples")
for K in (2, 4):
    parts = []
    for k in range(K):
        dp, d = make(8 // K)
        parts.append((dp, d, torch.cuda.Stream()))
    one = rate(parts[:1])
    seq = rate([(p[0], p[1], cur) for p in parts])
    con = rate(parts)
    print(f"{prec}: B={8 // K} alone {one:7.3f} ms;  {K} x B={8 // K} one stream {seq:7.3f} ms;  {K} x B={8 // K} on {K} streams {con:7.3f} ms per 8 samples")
