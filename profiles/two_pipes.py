"""Feasibility: two independent half-batch training pipelines on two streams vs one full-batch pipeline (timing only:
two separate models, no loss coupling).  python profiles/two_pipes.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bsms_gnn_amd as eng
from bench import build_workload, data_tuple, make_cfg

def make(batch, seed):
    wl = build_workload("airfoil", batch, "cuda", seed=seed)
    torch.manual_seed(0)
    sim = eng.BSMS_Simulator(make_cfg(wl["cfg"])).cuda()
    data = data_tuple(wl)
    sim(data, True, True)
    dp = eng.DataParallel(sim)
    return dp, data

def rate(pipes, steps=60, warm=10):
    streams = [torch.cuda.Stream() for _ in pipes]
    def one():
        for (dp, data), s in zip(pipes, streams):
            with torch.cuda.stream(s):
                dp.step_loss_backward(data, True)
    for _ in range(warm): one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): one()
    torch.cuda.synchronize()
    return steps / (time.perf_counter() - t0)

full = [make(8, 0)]
print("one pipeline, batch 8        : %.1f steps/s" % rate(full))
halves = [make(4, 0), make(4, 1)]
print("two pipelines, batch 4 each  : %.1f steps/s (of batch 8)" % rate(halves))
quarters = [make(2, i) for i in range(4)]
print("four pipelines, batch 2 each : %.1f steps/s (of batch 8)" % rate(quarters))
print("one pipeline, batch 4        : %.1f steps/s" % rate([halves[0]]))
