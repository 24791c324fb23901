"""Aggregation kernel per mesh level: time (HIP events per launch, warm and after flushing the caches) vs bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bsms_gnn_amd as eng
from bsms_gnn_amd import _abi
from bench import build_workload
wl = build_workload("airfoil", 8, "cuda")
flush = torch.empty(512 * 1024 * 1024 // 4, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for lvl, (n, e) in enumerate(wl["levels"]):
    g = wl["m_gs"][lvl][0]
    plan = eng.plan_for(g, n)
    x = torch.randn(8, e, 128, device="cuda")
    out = torch.empty(8, n, 128, device="cuda")
    def run():
        _abi.check(_abi.lib().bsms_segment_sum_fwd(plan.handle, x.data_ptr(), 8, 128, 0, out.data_ptr(), s), "seg")
    for _ in range(3): run()
    res = {}
    for mode in ("warm", "flushed"):
        ts = []
        for _ in range(10):
            if mode == "flushed": flush.zero_()
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts.sort(); res[mode] = ts[len(ts) // 2]
    mb = (8 * e + 8 * n) * 512 / 1e6
    print(f"level {lvl}: N={n} E={e}  {mb:6.1f} MB  warm {res['warm']:6.1f} us ({mb / res['warm']:.2f} TB/s)  flushed {res['flushed']:6.1f} us ({mb / res['flushed']:.2f} TB/s)")
