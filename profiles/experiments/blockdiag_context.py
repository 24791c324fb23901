import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench, bsms_gnn_amd as eng
mode = sys.argv[1]
def blockdiag():
    wl = bench.build_workload("cylinder", 8, "cuda")
    torch.manual_seed(0)
    sim = eng.BSMS_Simulator(bench.make_cfg(wl["cfg"])).cuda()
    data = bench.build_blockdiag_workload("cylinder", 8, "cuda")["data"]
    sim(data, False, True)
    dp = eng.DataParallel(sim)
    return bench.timed_steps(lambda: dp.step_loss_backward(data, False), 10, 30)[:2]
def dense():
    wl = bench.build_workload("cylinder", 8, "cuda")
    torch.manual_seed(0)
    sim = eng.BSMS_Simulator(bench.make_cfg(wl["cfg"])).cuda()
    data = bench.data_tuple(wl)
    sim(data, True, True)
    dp = eng.DataParallel(sim)
    return bench.timed_steps(lambda: dp.step_loss_backward(data, True), 10, 30)[:2]
if mode == "dense_first":
    print("dense", dense()); gc.collect(); torch.cuda.empty_cache()
    print("blockdiag after dense", blockdiag())
elif mode == "big_first":
    x = torch.empty(8 << 30, dtype=torch.uint8, device="cuda"); del x
    print("blockdiag after an 8 GB allocation (kept in the caching allocator)", blockdiag())
elif mode == "twice":
    print("blockdiag", blockdiag()); gc.collect(); torch.cuda.empty_cache()
    print("blockdiag again", blockdiag())
elif mode in ("airfoil_alone", "airfoil_after_cyl", "airfoil_after_blockdiag"):
    def airfoil():
        wl = bench.build_workload("airfoil", 8, "cuda")
        torch.manual_seed(0)
        sim = eng.BSMS_Simulator(bench.make_cfg(wl["cfg"])).cuda()
        data = bench.data_tuple(wl)
        sim(data, True, True)
        dp = eng.DataParallel(sim)
        return bench.timed_steps(lambda: dp.step_loss_backward(data, True), 20, 100)[:2]
    if mode == "airfoil_after_cyl":
        print("dense cylinder", dense()); gc.collect(); torch.cuda.empty_cache()
    if mode == "airfoil_after_blockdiag":
        print("blockdiag", blockdiag()); gc.collect(); torch.cuda.empty_cache()
    print(mode, airfoil())
elif mode == "lanes_first":
    torch.zeros(1, device="cuda")
    eng._abi.check(eng._abi.lib().bsms_side_lanes_join(torch.cuda.current_stream().cuda_stream), "join")   # creates both side lanes NOW
    print("blockdiag with the lanes created before anything else", blockdiag())
elif mode == "pool_first":
    s0 = torch.cuda.Stream()   # torch creates its pool of 32 streams
    print("dense after torch's stream pool", dense())
else:
    print("blockdiag alone", blockdiag())
