"""Phase stamps (s_memtime, wave 0) of the pipelined edge forward kernel k_edge_fwd, experiment build only
(BSMS_EXPERIMENTS=1 python bsms-gnn_amd/build.py --force):  BSMS_EDGE_RB=1|2 python profiles/edge_timeline.py
[0] tile start, [1] gathers issued, [2] input stage done, [3..5] MFMA stages done, [6] LayerNorm + message store issued."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bsms_gnn_amd as eng
from bench import build_workload
wl = build_workload("airfoil", 8, "cuda")
raw = ctypes.CDLL(eng._abi.LIB_PATH)
raw.bsms_debug_set_timing.argtypes = [ctypes.c_void_p]
rb = int(os.environ.get("BSMS_EDGE_RB", "2"))
for lvl in (0, 3):
    n0, e0 = wl["levels"][lvl]
    g0 = wl["m_gs"][lvl][0]
    plan = eng.plan_for(g0, n0)
    gmp = eng.GMP(128, 3, 2).cuda()
    pos = torch.rand(8, n0, 2, device="cuda")
    for grad in (True, False):
        x = torch.randn(8, n0, 128, device="cuda", requires_grad=grad)
        ntile = (8 * e0 + 63) // 64
        buf = torch.zeros(ntile * 16, dtype=torch.int64, device="cuda")
        ctx = torch.enable_grad() if grad else torch.no_grad()   # inference = the no-save kernel (stores only the messages)
        with ctx:
            for _ in range(3):
                gmp(x, g0, pos, plan=plan)
            raw.bsms_debug_set_timing(buf.data_ptr())
            gmp(x, g0, pos, plan=plan)
        torch.cuda.synchronize()
        raw.bsms_debug_set_timing(None)
        full = buf.cpu().numpy().reshape(ntile, 16).astype(np.float64)
        full = full[(full[:, :7] > 0).all(axis=1)]
        t = full[:, :7]
        if full[:, 11].max() > 0: print(f"  cycles of the stamped wave at the 12 chunk barriers: median {np.median(full[:, 11]):.0f}  p10 {np.percentile(full[:, 11], 10):.0f}  p90 {np.percentile(full[:, 11], 90):.0f}")
        if (full[:, 15] > 0).all():
            print(f"  shader clock over a tile (s_memtime / s_memrealtime): {np.median((t[:, 6] - t[:, 0]) / (full[:, 15] - full[:, 14])) * 100:.0f} MHz")
        d = np.diff(t, axis=1)
        names = ["issue gathers", "input stage (wait + fiber + relu)", "stage 0", "stage 1", "stage 2", "LayerNorm + y store"]
        span = t[:, 6].max() - t[:, 0].min()
        print(f"level {lvl} RB mode {rb} {'training' if grad else 'inference'}: {len(t)} tiles stamped, kernel span {span:.0f} ticks (s_memtime, 100 MHz?)")
        life = t[:, 6] - t[:, 0]
        print(f"  tile life median {np.median(life):.0f} p90 {np.percentile(life, 90):.0f}; sum of lives / span = {life.sum() / span:.1f} tiles in flight")
        for k, nm in enumerate(names):
            print(f"    {nm:36s} median {np.median(d[:, k]):8.0f}  p10 {np.percentile(d[:, k], 10):8.0f}  p90 {np.percentile(d[:, k], 90):8.0f}")
