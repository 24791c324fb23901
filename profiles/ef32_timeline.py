"""Phase stamps (s_memtime) of the fp32 fused edge backward k_edge_fused32_bwd (efuse32.hip, round 6: no recompute): chain wave 0 of every workgroup;
experiment build only, BSMS_EDGE_FUSED_F32=1.   python profiles/ef32_timeline.py [level]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bsms_gnn_amd as eng
from bench import build_workload
wl = build_workload("airfoil", 8, "cuda")
raw = ctypes.CDLL(eng._abi.LIB_PATH)
raw.bsms_debug_set_timing.argtypes = [ctypes.c_void_p]
names = ["previous g0 stores + locate + loads issued", "loads arrive + LayerNorm bwd", "row max + publish + barrier X + hand-over 3 + db", "dgrad 3 (4 chunk periods)", "mask",
         "row max + publish + X + hand-over 2 + db", "dgrad 2", "mask", "row max + publish + X + hand-over 1 + db", "dgrad 1", "mask", "g0 copy"]
for lvl in [int(a) for a in sys.argv[1:]] or [0, 3]:
    n0, e0 = wl["levels"][lvl]
    g0 = wl["m_gs"][lvl][0]
    net = eng.BSGMP(0, 128, 3, 2).cuda()
    pos = torch.rand(8, n0, 2, device="cuda")
    x = torch.randn(8, n0, 128, device="cuda", requires_grad=True)
    ntile = (8 * e0 + 63) // 64
    buf = torch.zeros(ntile * 16, dtype=torch.int64, device="cuda")
    for _ in range(3):
        net(x, [], [g0], pos).square().mean().backward()
    y = net(x, [], [g0], pos).square().mean()
    torch.cuda.synchronize()
    raw.bsms_debug_set_timing(buf.data_ptr())
    y.backward()
    torch.cuda.synchronize()
    raw.bsms_debug_set_timing(None)
    t = buf.cpu().numpy().reshape(ntile, 16).astype(np.float64)[:, :13]
    t = t[(t > 0).all(axis=1)]
    d = np.diff(t, axis=1)
    life = t[:, 12] - t[:, 0]
    print(f"level {lvl}: {ntile} tiles, {len(t)} stamped; s_memtime ticks; tile life median {np.median(life):.0f} p90 {np.percentile(life, 90):.0f}; launch span {t[:, 12].max() - t[:, 0].min():.0f}")
    for k, nm in enumerate(names):
        print(f"    {nm:52s} median {np.median(d[:, k]):7.0f}  p10 {np.percentile(d[:, k], 10):7.0f}  p90 {np.percentile(d[:, k], 90):7.0f}")
