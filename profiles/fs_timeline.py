"""Phase stamps (s_memtime, wave 0) of the feature-split node chain k_fs_fwd<IN_ROWS2, OUT_LN> (experiment build, flag 512):
[start, loads issued, rows + row max, pieces, pack0 (x half) MFMAs, x2 pieces, pack1 MFMAs, exch, pack2, exch, pack3, exch, pack4]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bsms_gnn_amd as eng
from bench import build_workload
raw = ctypes.CDLL(eng._abi.LIB_PATH)
raw.bsms_debug_set_timing.argtypes = [ctypes.c_void_p]
raw.bsms_debug_set_flags.argtypes = [ctypes.c_int]
names = ["issue loads", "rows arrive + row max", "publish pieces", "pack 0 (x) wait + MFMA", "publish x2", "pack 1 wait + MFMA + finish", "relu/max/publish",
         "pack 2", "relu/max/publish", "pack 3", "relu/max/publish", "pack 4"]
for B, lvl in ((1, 5), (1, 3), (1, 0)):
    wl = build_workload("airfoil", B, "cuda")
    n0, e0 = wl["levels"][lvl]
    g0 = wl["m_gs"][lvl][0]
    plan = eng.plan_for(g0, n0)
    gmp = eng.GMP(128, 3, 2).cuda()
    x = torch.randn(B, n0, 128, device="cuda")
    pos = torch.rand(B, n0, 2, device="cuda")
    ntile = (B * n0 + 15) // 16 + 8
    buf = torch.zeros(ntile * 16, dtype=torch.int64, device="cuda")
    with torch.no_grad():
        for rep in range(2):          # rep 0: packs cold in this kernel's view; rep 1: the same call again
            for _ in range(3 if rep == 0 else 0):
                gmp(x, g0, pos, plan=plan)
            buf.zero_()
            raw.bsms_debug_set_flags(512)
            raw.bsms_debug_set_timing(buf.data_ptr())
            gmp(x, g0, pos, plan=plan)
            torch.cuda.synchronize()
            raw.bsms_debug_set_timing(None)
            raw.bsms_debug_set_flags(0)
            full = buf.cpu().numpy().reshape(ntile, 16).astype(np.float64)
            t = full[full[:, 0] > 0][:, :13]
            d = np.diff(t, axis=1)
            print(f"\nB={B} level {lvl} ({B * n0} rows, {len(t)} tiles) rep {rep}: wave-0 life {np.median(t[:, 12] - t[:, 0]):.0f} cycles")
            for k, nm in enumerate(names):
                print(f"    {nm:34s} {np.median(d[:, k]):8.0f}  (p90 {np.percentile(d[:, k], 90):8.0f})")
