"""Phase stamps (s_memtime) of the fused edge backward k_edge_fused_bwd (efuse.hip): chain wave 0 and gradient wave 4 of every
workgroup; experiment build only (BSMS_EXPERIMENTS=1 python bsms-gnn_amd/build.py --force).   python profiles/ef_timeline.py [level]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bsms_gnn_amd as eng
from bench import build_workload
wl = build_workload("airfoil", 8, "cuda")
raw = ctypes.CDLL(eng._abi.LIB_PATH)
raw.bsms_debug_set_timing.argtypes = [ctypes.c_void_p]
names = ["issue dy/y loads + locate next + a0 (waits Ps/Pd)", "g0 stores 0-1 + fwd Linear 1 + relu/pack", "g0 stores 2-3 + fwd Linear 2 + relu/pack",
         "g0 stores 4-5 + LayerNorm bwd (waits dy/y) + g0 stores 6-7 + next Ps/Pd issue + pack",
         "dgrad 3 chunks 0-1", "barrier X3", "stage rows + chunk 2", "barrier Y3", "chunk 3 + mask + pack + dgrad 2 chunks 0-1", "barrier X2", "stage rows + chunk 2", "barrier Y2",
         "chunk 3 + mask + pack + dgrad 1 chunks 0-1", "barrier X1", "stage rows + chunk 2", "barrier Y1", "chunk 3 + mask", "gmax + pack g0"]
for lvl in [int(a) for a in sys.argv[1:]] or [0, 3]:
    n0, e0 = wl["levels"][lvl]
    g0 = wl["m_gs"][lvl][0]
    net = eng.BSGMP(0, 128, 3, 2).cuda()
    net.precision = "bf16"
    pos = torch.rand(8, n0, 2, device="cuda")
    x = torch.randn(8, n0, 128, device="cuda", requires_grad=True)
    ntile = (8 * e0 + 63) // 64
    buf = torch.zeros(ntile * 40, dtype=torch.int64, device="cuda")
    for _ in range(3):
        net(x, [], [g0], pos).square().mean().backward()
    y = net(x, [], [g0], pos).square().mean()
    torch.cuda.synchronize()
    raw.bsms_debug_set_timing(buf.data_ptr())
    y.backward()
    torch.cuda.synchronize()
    raw.bsms_debug_set_timing(None)
    full = buf.cpu().numpy().reshape(ntile, 40).astype(np.float64)
    t = full[:, :19]
    ok = (t > 0).all(axis=1)
    t = t[ok]
    d = np.diff(t, axis=1)
    life = t[:, 18] - t[:, 0]
    span = t[:, 18].max() - t[:, 0].min()
    print(f"level {lvl}: {ntile} tiles, {len(t)} stamped; s_memtime ticks (~ shader clock: 18.5k ticks per tile = 126-150 us per launch of 15.3 tiles per workgroup); tile life median {np.median(life):.0f} p90 {np.percentile(life, 90):.0f} ticks")
    for k, nm in enumerate(names):
        print(f"    {nm:58s} median {np.median(d[:, k]):7.0f}  p10 {np.percentile(d[:, k], 10):7.0f}  p90 {np.percentile(d[:, k], 90):7.0f}")
    gwt = full[ok][:, 24:36]
    if not (gwt > 0).any():   # gradient-wave stamps need a build with -DEFV_GSTAMP (they perturb the kernel: DESIGN.md 4.9)
        continue
    gd = np.diff(gwt, axis=1)
    gn = ["X wait", "Y wait", "consume (16 tr reads x2, 32 MFMA, colsum)", "(to next)"] * 3
    print("  gradient wave 4:")
    for k in range(11):
        print(f"    l={3 - k // 4} {gn[k]:48s} median {np.median(gd[:, k]):7.0f}  p90 {np.percentile(gd[:, k], 90):7.0f}")
