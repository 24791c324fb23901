"""Where does a k_wgrad step go?  s_memtime stamps of waves 0 (stage first) and 4 (MFMA first) of every workgroup:
per step [start, after first half, after second half] (the barrier is the gap to the next start)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bsms_gnn_amd as eng
raw = ctypes.CDLL(eng._abi.LIB_PATH)
raw.bsms_debug_set_wgrad_timing.argtypes = [ctypes.c_void_p]
R = 250880
mlp = eng.MLP(128, 128, 128, 2, True).cuda()
x = torch.randn(R, 128, device="cuda", requires_grad=True)
for _ in range(2):
    mlp(x).sum().backward()
buf = torch.zeros(1024 * 2 * 64, dtype=torch.int64, device="cuda")
raw.bsms_debug_set_wgrad_timing(buf.data_ptr())
y = mlp(x)
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); y.sum().backward(); t1.record(); torch.cuda.synchronize()
raw.bsms_debug_set_wgrad_timing(None)
t = buf.cpu().numpy().reshape(1024, 2, 64).astype(np.float64)
used = t[:, 0, 0] > 0
t = t[used]
print(f"{used.sum()} workgroups stamped; whole backward {t0.elapsed_time(t1) * 1e3:.0f} us")
for w, name in ((0, "wave 0 (stage, then MFMA)"), (1, "wave 4 (MFMA, then stage)")):
    tw = t[:, w, :63].reshape(len(t), 21, 3)
    ok = (tw > 0).all(axis=(1, 2))
    tw = tw[ok]
    first = tw[:, :, 1] - tw[:, :, 0]
    second = tw[:, :, 2] - tw[:, :, 1]
    gap = tw[:, 1:, 0] - tw[:, :-1, 2]
    step = tw[:, 1:, 0] - tw[:, :-1, 0]
    print(f"{name}: median cycles per step {np.median(step):.0f}; first half {np.median(first):.0f}, second half {np.median(second):.0f}, barrier wait {np.median(gap):.0f}")
    print("   per step index (median step cycles):", [int(np.median(step[:, i])) for i in range(0, 20, 2)])
