"""Where does a tile's lifetime go in the edge forward chain?  s_memtime stamps of wave 0 of every workgroup:
[start, input stage done, first barrier passed, stage0 done, stage1 done, stage2 done, end-of-loop]."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bsms_gnn_amd as eng
from bench import build_workload
wl = build_workload("airfoil", 8, "cuda")
raw = ctypes.CDLL(eng._abi.LIB_PATH)
raw.bsms_debug_set_timing.argtypes = [ctypes.c_void_p]
for lvl in (0, 4):
    n0, e0 = wl["levels"][lvl]
    g0 = wl["m_gs"][lvl][0]
    plan = eng.plan_for(g0, n0)
    gmp = eng.GMP(128, 3, 2).cuda()
    x = torch.randn(8, n0, 128, device="cuda", requires_grad=True)
    pos = torch.rand(8, n0, 2, device="cuda")
    nwg = (8 * e0 + 15) // 16                              # upper bound; unstamped rows are dropped below
    buf = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
    for _ in range(3):
        gmp(x, g0, pos, plan=plan)
    raw.bsms_debug_set_timing(buf.data_ptr())
    gmp(x, g0, pos, plan=plan)
    torch.cuda.synchronize()
    raw.bsms_debug_set_timing(None)
    full = buf.cpu().numpy().reshape(nwg, 16).astype(np.float64)
    full = full[full[:, 14] > 0]
    nwg = len(full)
    real = full[:, 15] - full[:, 14]                       # 100 MHz constant clock
    core = full[:, 6] - full[:, 0]
    print(f"level {lvl}: shader clock during the kernel = {np.median(core / real) * 100:.0f} MHz "
          f"(s_memtime / s_memrealtime, median over workgroups); tile life {np.median(real) / 100:.1f} us")
    t = full[:, :7]
    d = np.diff(t, axis=1)
    names = ["input stage (gather)", "first barrier wait", "stage 0", "stage 1", "stage 2", "loop exit"]
    ok = (t > 0).all(axis=1)
    t, d = t[ok], d[ok]
    life = t[:, 6] - t[:, 0]
    span = t[:, 6].max() - t[:, 0].min()
    print(f"level {lvl}: {nwg} workgroups ({ok.sum()} stamped), span {span:.0f} ticks, sum of tile lives / span = "
          f"{life.sum() / span:.1f} concurrent tiles on the chip = {life.sum() / span / 256:.2f} per CU")
    starts = np.sort(t[:, 0] - t[:, 0].min())
    print("  start times (ticks) of workgroup #0, #255, #1023, #2047, last:", [int(starts[min(i, len(starts) - 1)]) for i in (0, 255, 1023, 2047, len(starts) - 1)])
    print("  median tile life %.0f ticks; phases (median / p90):" % np.median(life))
    print(f"    cycles waiting at the 12 chunk barriers (median / p90): {np.median(full[ok][:, 11]):9.0f} {np.percentile(full[ok][:, 11], 90):9.0f}")
    for k, nm in enumerate(names):
        print(f"    {nm:24s} {np.median(d[:, k]):9.0f} {np.percentile(d[:, k], 90):9.0f}")

    # residency per CU from the chip-wide constant clock (10 ns ticks) and HW_ID / XCC_ID of wave 0
    hw = full[:, 13].astype(np.int64)
    xcc, hwid = (hw >> 32) & 0xF, hw & 0xFFFFFFFF
    cu, sh, se = (hwid >> 8) & 0xF, (hwid >> 12) & 0x1, (hwid >> 13) & 0x7
    simd = (hwid >> 4) & 0x3
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    st, en_loop, en_all = full[:, 14], full[:, 15], full[:, 12]
    t0, t1 = st.min(), en_all.max()
    print(f"  kernel span {(t1 - t0) / 100:.1f} us; distinct CUs seen {len(np.unique(key))}; wave-0 SIMD histogram {np.bincount(simd, minlength=4)}")
    print(f"  tile life to loop exit {np.median(en_loop - st) / 100:.1f} us, to last store ack {np.median(en_all - st) / 100:.1f} us")
    print(f"  average resident tiles per CU (to last store ack): {(en_all - st).sum() / (t1 - t0) / len(np.unique(key)):.2f}")
    per_cu = np.bincount(np.unique(key, return_inverse=True)[1])
    print(f"  tiles per CU: min {per_cu.min()} median {np.median(per_cu):.0f} max {per_cu.max()}")
    # concurrency histogram on one busy CU, sampled mid-kernel
    k0 = np.unique(key)[len(np.unique(key)) // 2]
    m = key == k0
    ts = np.linspace(t0 + 0.2 * (t1 - t0), t0 + 0.8 * (t1 - t0), 200)
    conc = [(np.logical_and(st[m] <= t, en_all[m] > t)).sum() for t in ts]
    print(f"  CU {k0}: resident tiles mid-kernel histogram {np.bincount(conc)}")
    gaps = np.sort(st[m])
    print(f"  CU {k0}: first 12 start times (us) {[round((g - t0) / 100, 1) for g in gaps[:12]]}")
    uk, inv = np.unique(key, return_inverse=True)
    early = np.bincount(inv, weights=(st - t0 < 300).astype(np.float64))
    print(f"  workgroups started in the first 3 us, per CU: histogram {np.bincount(early.astype(int))}")
    peak = []
    for c in range(len(uk)):
        m = inv == c
        ev = np.concatenate([np.stack([st[m], np.ones(m.sum())], 1), np.stack([en_all[m], -np.ones(m.sum())], 1)])
        ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
        peak.append(int(np.cumsum(ev[:, 1]).max()))
    print(f"  peak resident (wave-0 start .. last store ack) per CU: histogram {np.bincount(peak)}")
