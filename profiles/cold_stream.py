"""What does the chip deliver on COLD streams of the aggregation's size?  Rotating 128 MB buffers (6 of them: 768 MB, the
memory-side cache holds 256 MB), HIP events per launch: torch copy (read 128 MB + write 128 MB), torch sum (read only),
and the L0 aggregation itself (read 128 MB + write 21 MB).  python profiles/cold_stream.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bsms_gnn_amd as eng
from bsms_gnn_amd import _abi
from bench import build_workload
wl = build_workload("airfoil", 8, "cuda")
n, e = wl["levels"][0]
plan = eng.plan_for(wl["m_gs"][0][0], n)
xs = [torch.randn(8, e, 128, device="cuda") for _ in range(6)]
ys = [torch.empty(8, e, 128, device="cuda") for _ in range(6)]
out = torch.empty(8, n, 128, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def timed(fn, iters=60):
    ts = []
    for i in range(iters + 6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(i); b.record(); torch.cuda.synchronize()
        if i >= 6: ts.append(a.elapsed_time(b) * 1e3)
    ts.sort(); return ts[len(ts) // 2], ts[len(ts) // 10]
mb = 8 * e * 512 / 1e6
t, t10 = timed(lambda i: ys[i % 6].copy_(xs[(i + 3) % 6]))
print(f"copy   {2 * mb:6.1f} MB cold: median {t:5.1f} us = {2 * mb / t:.2f} TB/s (p10 {t10:.1f})")
t, t10 = timed(lambda i: torch.sum(xs[i % 6], dim=(0, 1), out=out[0, 0]) if False else xs[i % 6].sum())
print(f"sum    {mb:6.1f} MB cold: median {t:5.1f} us = {mb / t:.2f} TB/s (p10 {t10:.1f})")
def agg(i):
    _abi.check(_abi.lib().bsms_segment_sum_fwd(plan.handle, xs[i % 6].data_ptr(), 8, 128, 1, out.data_ptr(), s), "seg")
t, t10 = timed(agg)
tot = (8 * e + 8 * n) * 512 / 1e6
print(f"aggregation {tot:6.1f} MB cold: median {t:5.1f} us = {tot / t:.2f} TB/s (p10 {t10:.1f})")
t, t10 = timed(lambda i: _abi.check(_abi.lib().bsms_segment_sum_fwd(plan.handle, xs[0].data_ptr(), 8, 128, 1, out.data_ptr(), s), "seg"))
print(f"aggregation {tot:6.1f} MB warm: median {t:5.1f} us = {tot / t:.2f} TB/s (p10 {t10:.1f})")
