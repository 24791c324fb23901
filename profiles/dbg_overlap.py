"""Where does the overlapped all-reduce path spend host time under gloo (2 ranks on one GPU)?  torchrun-less: spawned ranks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist, torch.multiprocessing as mp

def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import bench, bsms_gnn_amd as eng
    from bsms_gnn_amd import step as st
    wl = bench.build_workload("cylinder", 8, "cuda", seed=rank)
    torch.manual_seed(0)
    sim = eng.BSMS_Simulator(bench.make_cfg(wl["cfg"])).cuda()
    data = bench.data_tuple(wl)
    sim(data, True, True)
    dp = eng.DataParallel(sim)
    fs = dp.fused
    orig_ar = dist.all_reduce
    times = []
    def timed_ar(t, *a, **k):
        t0 = time.perf_counter(); r = orig_ar(t, *a, **k); times.append(("all_reduce call", t.numel(), time.perf_counter() - t0)); return r
    dist.all_reduce = timed_ar
    st.dist.all_reduce = timed_ar
    for it in range(6):
        times.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dp.step_loss_backward(data, True)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        if rank == 0 and it >= 3:
            print(f"step {it}: host {1e3 * (t1 - t0):.1f} ms, +sync {1e3 * (t2 - t1):.1f} ms; " + ", ".join(f"{n}:{1e3 * d:.1f}ms" for _, n, d in times))
    dist.destroy_process_group()

if __name__ == "__main__":
    mp.start_processes(worker, args=(2, 29911), nprocs=2, join=True, start_method="spawn")
