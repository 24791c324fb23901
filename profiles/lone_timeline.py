"""Where does a single-round ("LONE") node-chain launch spend its ~19 us?  s_memtime stamps of wave 0 of every workgroup of
k_chain_fwd<8, IN_ROWS2, OUT_LN, timing, -, LONE> (experiment build + debug flag 512), node MLP of a GMP block:
[start, loads issued, =, x arrived, x2 arrived, stage 0 first half, stage 0, stage 1, stage 2, stage 3, loop exit] + store ack.
   BSMS_EXPERIMENTS build as libbsms_hip.so;  python profiles/lone_timeline.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bsms_gnn_amd as eng
from bench import build_workload
raw = ctypes.CDLL(eng._abi.LIB_PATH)
raw.bsms_debug_set_timing.argtypes = [ctypes.c_void_p]
raw.bsms_debug_set_flags.argtypes = [ctypes.c_int]
names = ["issue loads", "-", "x arrives", "x2 arrives (2nd load + amax)", "stage 0 a (x half)", "stage 0 b (x2 half, 3rd load)", "stage 1", "stage 2", "stage 3", "loop exit"]
for B, lvl, train in ((1, 5, False), (1, 3, False), (1, 0, False), (8, 5, False), (1, 3, True), (8, 3, True)):
    wl = build_workload("airfoil", B, "cuda")
    n0, e0 = wl["levels"][lvl]
    g0 = wl["m_gs"][lvl][0]
    plan = eng.plan_for(g0, n0)
    gmp = eng.GMP(128, 3, 2).cuda()
    x = torch.randn(B, n0, 128, device="cuda", requires_grad=train)
    pos = torch.rand(B, n0, 2, device="cuda")
    ntile = (B * n0 + 63) // 64 + 8
    buf = torch.zeros(ntile * 16, dtype=torch.int64, device="cuda")
    ctx = torch.enable_grad() if train else torch.no_grad()
    with ctx:
        for _ in range(3):
            gmp(x, g0, pos, plan=plan)
        raw.bsms_debug_set_flags(512)
        raw.bsms_debug_set_timing(buf.data_ptr())
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); gmp(x, g0, pos, plan=plan); b.record()
        torch.cuda.synchronize()
        raw.bsms_debug_set_timing(None)
        raw.bsms_debug_set_flags(0)
    full = buf.cpu().numpy().reshape(ntile, 16).astype(np.float64)
    full = full[full[:, 14] > 0]
    t = full[:, :11]
    ok = (t > 0).all(axis=1)
    t = t[ok]
    d = np.diff(t, axis=1)
    real = full[ok][:, 15] - full[ok][:, 14]
    clk = np.median((t[:, 10] - t[:, 0]) / real) * 100
    print(f"\nB={B} level {lvl} ({B * n0} rows, {len(full)} tiles stamped) {'training' if train else 'inference'}: whole GMP forward {a.elapsed_time(b) * 1e3:.1f} us; "
          f"shader clock {clk:.0f} MHz; wave-0 life to loop exit {np.median(t[:, 10] - t[:, 0]):.0f} cycles = {np.median(real) / 100:.2f} us, "
          f"to last store ack {np.median(full[ok][:, 12] - full[ok][:, 14]) / 100:.2f} us; kernel span {(full[:, 12].max() - full[:, 14].min()) / 100:.2f} us")
    print(f"    cycles waiting at the 12 chunk barriers of stages 1-3 (s_waitcnt lgkmcnt(0) + s_barrier): median {np.median(full[ok][:, 11]):.0f} (p90 {np.percentile(full[ok][:, 11], 90):.0f})")
    for k, nm in enumerate(names):
        print(f"    {nm:34s} {np.median(d[:, k]):8.0f} cycles  (p90 {np.percentile(d[:, k], 90):8.0f})")
