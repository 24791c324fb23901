#!/usr/bin/env python3
"""Benchmark of the BSMS-GNN hot path on MI355X (contract: see the task prompt / DESIGN.md).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...          (no launcher: bench.py starts its own N ranks, one per GPU, backend nccl = RCCL;
                                         on a box with fewer than N GPUs the ranks share devices over gloo -- a functional
                                         check of the N > 1 path, flagged as such on the JSON line)

One "step" = BSMS_Simulator forward (warmup=False) + masked-RMSE loss + backward with every parameter
gradient materialised (N > 1: including the RCCL gradient all-reduce), on the airfoil-like synthetic
workload of BASELINE.json (5233-node Delaunay mesh, 5 bi-stride levels, D=128, batch 8 PER GPU, fp32),
inputs resident in HBM.  `value` = batch-8 steps per second summed over all ranks (weak scaling).

Extra objects on the JSON line (rank 0, N = 1 only):
  roofline      the L0 edge-aggregation kernel (HBM bound), timed with HIP events on its stream
  roofline_mfma the edge-MLP chain kernels (MFMA bound), same method
  cpu_baseline  the CPU oracle (op-for-op restatement of the reference's PyTorch path) on the host cores
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {  # SURVEY.md section 8(d); BASELINE.json configs[1], [2]/[3], [4]
    "cylinder": dict(nodes=1885, levels=4, out_dim=2, pos_dim=2, latent=128, mesh_seed=0),
    "airfoil": dict(nodes=5233, levels=5, out_dim=3, pos_dim=2, latent=128, mesh_seed=0),
    # inflating-surface stand-in: 16384 nodes on z = sin(3u) cos(3v), 3-D positions, 6 levels, D=256.  Mesh seed 1:
    # with seed 0 the level-5 graph is complete, its BFS keeps ONE node and level 6 degenerates to N=1, E=0
    # (SURVEY.md section 8 table); seed 1 bottoms out at 88 nodes / 4796 edges (validated by validate_hierarchy).
    "surface": dict(nodes=16384, levels=6, out_dim=3, pos_dim=3, latent=256, mesh_seed=1),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: f32-input MFMA = f32 vector rate
MFMA_BF16_PEAK_TF = 2516.6  # dense bf16 MFMA: 256 CUs x 4 SIMDs x 1024 flop/cycle x 2.4 GHz
MFMA_SPLIT_PEAK_TF = MFMA_BF16_PEAK_TF / 3  # fp32 by two-way fp16 split = three f16 products per fp32 MAC (f16 and bf16 MFMA peaks are equal)


def mesh_points(kind, seed=None):
    """Node positions [N,p] (fp64) and the triangulation of the synthetic stand-in mesh."""
    from scipy.spatial import Delaunay
    w = WORKLOADS[kind]
    uv = np.random.default_rng(w["mesh_seed"] if seed is None else seed).random((w["nodes"], 2))
    cells = Delaunay(uv).simplices.astype(np.int64)
    if w["pos_dim"] == 3:
        uv = np.concatenate([uv, (np.sin(3 * uv[:, 0]) * np.cos(3 * uv[:, 1]))[:, None]], 1)
    return uv, cells


def validate_hierarchy(kind, m_es, m_ids):
    """A usable hierarchy: every level has >= 2 nodes, >= 1 edge, every node an out-edge (degree() quirk,
    utils/basic.py:305) and the level sizes chain (len(m_ids[l]) nodes at level l+1)."""
    n = WORKLOADS[kind]["nodes"]
    for l, e in enumerate(m_es):
        if n < 2 or e.shape[1] < 1:
            raise ValueError(f"{kind}: level {l} is degenerate (N={n}, E={e.shape[1]}); pick another mesh seed / depth")
        if int(e.max()) >= n or len(np.unique(e[0])) != n:
            raise ValueError(f"{kind}: level {l} has a node without an out-edge or an index out of range")
        if l < len(m_ids):
            n = len(m_ids[l])


def build_mesh(kind, seed=None):
    """Synthetic Delaunay stand-in for the dataset mesh + its bi-stride hierarchy (host, once)."""
    from bsms_gnn_amd.hierarchy import BistrideMultiLayerGraph, to_flat_edge
    w = WORKLOADS[kind]
    pts, cells = mesh_points(kind, seed)
    flat = to_flat_edge(cells, "tri")
    _, m_es, m_ids = BistrideMultiLayerGraph(flat, w["levels"], w["nodes"], pts).get_multi_layer_graphs()
    validate_hierarchy(kind, m_es, m_ids)
    return pts, m_es, m_ids


def strip_mesh(nx=327, ny=16, levels=7):
    """Structured, jittered triangle strip (nx*ny nodes).  A uniform random 5k-node Delaunay mesh collapses to one
    node at level 6, so the reference's DEFAULT depth (configs/model/airfoil.yaml:4 `unet_depth: 7`) needs a mesh
    with a large graph diameter, like the real (graded, elongated) airfoil mesh: 327 x 16 -> 32 nodes at level 7."""
    from bsms_gnn_amd.hierarchy import BistrideMultiLayerGraph, to_flat_edge
    xs, ys = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    idx = xs * ny + ys
    pts = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.float64) + 0.2 * np.random.default_rng(0).random((nx * ny, 2))
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    cells = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)]).astype(np.int64)
    w = dict(WORKLOADS["airfoil"], nodes=nx * ny, levels=levels)
    _, m_es, m_ids = BistrideMultiLayerGraph(to_flat_edge(cells, "tri"), levels, nx * ny, pts).get_multi_layer_graphs()
    return w, (pts / ny, m_es, m_ids)


def build_workload(kind, batch, device, seed=0, mesh=None, cfg=None):
    """Consistent-mesh batch exactly as the reference collates it (datasets/base.py:319-351 + default
    collate): node_in [B,N,C+p+1] = [state, mesh_pos, node_type], every m_gs[l] as [B,2,E_l].
    `mesh` = (pts, m_es, m_ids) with its own `cfg` dict overrides the named workload's mesh."""
    w = WORKLOADS[kind] if cfg is None else cfg
    pts, m_es, m_ids = build_mesh(kind) if mesh is None else mesh
    n, c = w["nodes"], w["out_dim"]
    gen = torch.Generator().manual_seed(seed)
    state = torch.randn(batch, n, c, generator=gen)
    target = torch.randn(batch, n, c, generator=gen)
    pos = torch.tensor(pts, dtype=torch.float32).unsqueeze(0).repeat(batch, 1, 1)
    node_in = torch.cat([state, pos, torch.zeros(batch, n, 1)], -1)
    mask = torch.ones(batch, n, 1)
    gs = [torch.tensor(e, dtype=torch.int64).unsqueeze(0).repeat(batch, 1, 1) for e in m_es]
    ids = [torch.tensor(i, dtype=torch.int64).unsqueeze(0).repeat(batch, 1) for i in m_ids]
    mv = lambda t: t.to(device)
    return dict(node_in=mv(node_in), target=mv(target), mask=mv(mask), m_gs=[mv(g) for g in gs], m_ids=[mv(i) for i in ids],
                levels=[(n if l == 0 else len(m_ids[l - 1]), m_es[l].shape[1]) for l in range(len(m_es))], cfg=w)


def build_blockdiag_workload(kind, batch, device):
    """Variable-mesh layout (the reference's cylinder_flow path, consistent_mesh: false): `batch` DIFFERENT meshes
    (Delaunay seeds 0..batch-1), collated into one block-diagonal graph per level (A15)."""
    import bsms_gnn_amd as eng
    w = WORKLOADS[kind]
    n, c = w["nodes"], w["out_dim"]
    gen = torch.Generator().manual_seed(0)
    samples = []
    for seed in range(batch):
        pts, m_es, m_ids = build_mesh(kind, seed=seed)
        state, target = torch.randn(n, c, generator=gen), torch.randn(n, c, generator=gen)
        x = torch.cat([state, torch.tensor(pts, dtype=torch.float32), torch.zeros(n, 1)], -1)
        sizes = [n] + [len(i) for i in m_ids]
        samples.append([eng.LevelData(torch.tensor(m_es[l]), sizes[l], face=torch.tensor(m_ids[l]) if l < w["levels"] else None,
                                      x=x if l == 0 else None, y=target if l == 0 else None,
                                      mask=torch.ones(n, 1) if l == 0 else None) for l in range(w["levels"] + 1)])
    batchd = [d.to(device) for d in eng.collate_variable_meshes(samples)]
    return dict(data=batchd, samples=samples, cfg=w,
                levels=[(d.num_nodes, int(d.edge_index.shape[1])) for d in batchd])


def make_cfg(w):
    from types import SimpleNamespace
    return SimpleNamespace(out_dim=w["out_dim"], latent_dim=w["latent"], hidden_layer=3, unet_depth=w["levels"],
                           pos_dim=w["pos_dim"], consistent_mesh=True, accumulation_steps=0)


def data_tuple(wl):
    return (wl["node_in"], wl["target"], wl["mask"], wl["m_gs"], wl["m_ids"])


def usable_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup quota (a 256-thread host often
    hands a container far fewer; 256 torch threads on a quota of a few cores is 30x slower than 8 threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _oracle_steps(kind, batch, threads, max_steps, budget_s, warm=1):
    """fwd + loss + bwd of the CPU oracle; returns (seconds per step: best after the warm-up step, steps run, loss)."""
    from oracle import bsms_oracle as ro
    torch.set_num_threads(threads)
    wl = build_workload(kind, batch, "cpu")
    torch.manual_seed(0)
    sim = ro.BSMS_Simulator(make_cfg(wl["cfg"]))
    data = data_tuple(wl)
    sim(data, True, True)
    times, t_start, loss = [], time.perf_counter(), None
    while len(times) < max_steps and (len(times) < 1 or time.perf_counter() - t_start < budget_s):
        sim.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        loss = ro.masked_rmse(sim(data, True, False), wl["target"], wl["mask"])
        loss.backward()
        times.append(time.perf_counter() - t0)
    best = min(times[warm:]) if len(times) > warm else min(times[1:]) if len(times) > 1 else times[0]   # the first step(s) double as warm-up when there is time
    return best, len(times), float(loss.detach())


def cpu_baseline(kind, batch, budget_s=30.0, steps=0):
    """The oracle (CPU restatement of the reference path, `kind: port`) on the host cores, bounded to about
    `budget_s` + 15 s of CPU work so that the default bench run stays within minutes (SURVEY.md section 8d asks for
    best-of-5: at 5-15 s per airfoil step that is minutes, so the sample is 1 warm-up + up to 3 timed steps):
      * all usable threads (affinity + cgroup quota, capped at 32: the reference's small per-level ops stop scaling):
        the SAME workload as the GPU line (same seed -> its loss must equal the GPU loss, checked by main());
      * 1 thread: the SAME batch-`batch` step on one thread (one warm-up step when it fits the budget, then one timed
        step: ~11 s each at airfoil size) -- measured, not scaled from a smaller batch."""
    threads = max(1, min(usable_cpus(), 32))
    if steps > 0:   # --cpu-baseline-steps K: SURVEY.md 8(d)'s protocol -- 2 warm-ups + best of K, however long it takes
        best, n, loss = _oracle_steps(kind, batch, threads, 2 + steps, 1e9, warm=2)
        one, n1, _ = _oracle_steps(kind, batch, 1, 2, 0.6 * budget_s)
        torch.set_num_threads(threads)
        return {"value": 1.0 / best, "unit": "steps/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
                "sample": f"{kind}-like B={batch} fwd+loss+bwd, 2 warm-up steps + best of {steps} (SURVEY.md 8(d) protocol), fp32, "
                          f"torch CPU {torch.__version__}, {threads} threads of {os.cpu_count()} logical CPUs",
                "ms_per_step": best * 1e3, "loss": loss,
                "one_thread": {"value": 1.0 / one, "unit": "steps/s", "cores": 1, "ms_per_step": one * 1e3,
                               "sample": f"the same B={batch} step on ONE thread, {n1} step(s)"}}
    best, n, loss = _oracle_steps(kind, batch, threads, 4, budget_s)
    one, n1, _ = _oracle_steps(kind, batch, 1, 2, 0.6 * budget_s)
    torch.set_num_threads(threads)
    return {"value": 1.0 / best, "unit": "steps/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"{kind}-like B={batch} fwd+loss+bwd, {n} step(s) within a {budget_s:.0f} s budget "
                      f"(best of the non-warm-up ones), fp32, torch CPU {torch.__version__}, "
                      f"{threads} threads of {os.cpu_count()} logical CPUs",
            "ms_per_step": best * 1e3, "loss": loss,
            "one_thread": {"value": 1.0 / one, "unit": "steps/s", "cores": 1, "ms_per_step": one * 1e3,
                           "sample": f"the same B={batch} step on ONE thread, {n1} step(s) "
                                     f"({'best of the non-warm-up ones' if n1 > 1 else 'no warm-up step fitted the budget'})"}}


def time_kernel(fn, iters=50, warm=5):
    """Average duration (ms) of ONE launch of `fn`, every launch bracketed by its OWN HIP event pair on the launching
    (current) stream.  This includes, per launch, the event packets and the dispatch / completion latency around the
    kernel (~3 us on MI355X: 27.9 us against the 24.75 us rocprofv3 reports for the same cold aggregation launches)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in pairs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in pairs) / iters


def time_kernel_stream(fn, iters=60, warm=5):
    """Average duration (ms) of one launch of `fn` from ONE HIP event pair around `iters` back-to-back launches on the
    launching stream, divided by `iters`.  The stream is in-order (every dispatch packet carries the barrier bit), so
    launch i + 1 starts only after launch i has drained: the figure is kernel duration + one dispatch gap, i.e. still an
    upper bound of what rocprofv3's kernel trace reports per dispatch, without the event packets of the per-launch form
    (rounds 1-3 used that form: it disagreed with the committed rocprofv3 averages by +13 %)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def roofline_objects(wl, batch, dtype="f32"):
    """HBM roofline of the L0 edge aggregation and MFMA roofline of the L0 edge-MLP forward chain.  dtype "bf16": the
    aggregation kernel of the bf16 precision (bf16 messages in, fp32 sums out; SURVEY.md 8(d): 75.1 MB at airfoil L0)."""
    import ctypes as C
    import bsms_gnn_amd as eng
    from bsms_gnn_amd import _abi
    from bsms_gnn_amd.ops import _stream
    n0, e0 = wl["levels"][0]
    D = wl["cfg"]["latent"]
    g0 = wl["m_gs"][0][0]
    plan = eng.plan_for(g0, n0)
    L = _abi.lib()
    bf = dtype != "f32"
    s = 2 if bf else 4
    algo = batch * e0 * D * s + batch * n0 * D * 4 + 4 * (n0 + 1) + 4 * e0   # SURVEY.md section 8(d); the sums stay fp32
    # COLD: rotate over enough message buffers that a launch never finds its input in the 256 MiB memory-side cache
    # (Infinity Cache): this is what the kernel sees inside the training step, where the messages were just streamed
    # out by the edge chain.  WARM: one buffer pair re-read (fits the cache) -- reported, but not the roofline claim.
    nbuf = max(3, int(np.ceil(640e6 / (batch * e0 * D * s))) + 1)
    msgs = [torch.randn(batch, e0, D, device="cuda", dtype=torch.bfloat16 if bf else torch.float32) for _ in range(nbuf)]
    outs = [torch.empty(batch, n0, D, device="cuda") for _ in range(nbuf)]
    state = {"i": 0}

    def launch(i):
        if bf:
            _abi.check(L.bsms_segment_sum_bf16(plan.handle, msgs[i].data_ptr(), batch, D, outs[i].data_ptr(), _stream()), "segment_sum_bf16")
        else:
            _abi.check(L.bsms_segment_sum_fwd(plan.handle, msgs[i].data_ptr(), batch, D, 1, outs[i].data_ptr(), _stream()), "segment_sum")

    def agg_cold():
        i = state["i"] = (state["i"] + 1) % nbuf
        launch(i)

    agg_warm = lambda: launch(0)
    ms = time_kernel_stream(agg_cold, iters=60)        # the claim: agrees with the rocprofv3 average of the same command (profiles/)
    ms_pair = time_kernel(agg_cold, iters=60)          # rounds 1-3: one event pair per launch (+ ~3 us of event / dispatch latency)
    ms_warm = time_kernel_stream(agg_warm, iters=50)
    # what the chip delivers on a plain device copy under the SAME cold rotation (read one message buffer, write another):
    # the practical ceiling for cold streams of this size, reported next to the claim (the claim stays against 8 TB/s)
    def copy_cold():
        i = state["i"] = (state["i"] + 1) % nbuf
        msgs[i].copy_(msgs[(i + nbuf // 2) % nbuf])
    ms_copy = time_kernel_stream(copy_cold, iters=30)
    copy_gbs = 2 * batch * e0 * D * s / (ms_copy * 1e-3) / 1e9
    del msgs, outs
    traffic, in_step, traffic_run = None, None, None
    try:   # HBM bytes per launch from the separate rocprofv3 --pmc passes (profiles/*_traffic.json)
        tj = json.load(open(os.path.join(ROOT, "profiles", "aggregation_traffic.json")))
        traffic, traffic_run = tj["hbm_bytes_per_launch"], tj.get("taken")
    except (OSError, KeyError, ValueError):
        pass
    try:   # the same kernel inside the profiled training step (profiles/summarize.py writes this next to the summary)
        in_step = json.load(open(os.path.join(ROOT, "profiles", "aggregation_in_step.json")))
    except (OSError, ValueError):
        pass
    gbs = lambda t_ms: algo / (t_ms * 1e-3) / 1e9
    if bf:
        traffic, in_step = None, None   # the PMC passes and the in-step profile were taken for the fp32 kernel
    roof = {"kernel": ("k_rowsum_bf16in (eight features per lane; L0 edge aggregation of the bf16 precision, bsms_segment_sum_bf16)" if bf else
                       "k_rowsum_v4<32,false,false,false> (L0 edge aggregation, bsms_segment_sum_fwd plan order)"),
            "bound": "hbm", "achieved": gbs(ms), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs(ms) / HBM_PEAK_GBS,
            "traffic": traffic, "algorithmic_bytes": algo, "avg_us": ms * 1e3,
            "method": f"cold: {nbuf} rotating message buffers ({nbuf * batch * e0 * D * s / 2**20:.0f} MiB > 256 MiB memory-side cache), "
                      "60 back-to-back launches on the launching (in-order) stream inside ONE HIP event pair, / 60",
            "avg_us_event_pair_per_launch": ms_pair * 1e3, "frac_event_pair_per_launch": gbs(ms_pair) / HBM_PEAK_GBS,
            "frac_cold": gbs(ms) / HBM_PEAK_GBS, "frac_warm": gbs(ms_warm) / HBM_PEAK_GBS, "avg_us_warm": ms_warm * 1e3,
            "frac_in_step": None if not in_step else in_step.get("frac"), "in_step": in_step,
            # `traffic`, `frac_in_step` and `in_step` are NOT measured by this run: PMC counters need their own rocprofv3 passes and
            # the in-step figure a kernel trace; they are read from the committed profiles of the same kernel
            "source": {"achieved / frac / avg_us / frac_warm / cold_device_copy": "measured in this run (HIP events)",
                       "traffic": None if traffic is None else f"committed profile profiles/aggregation_traffic.json (run {traffic_run}; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
                       "frac_in_step / in_step": None if not in_step else f"committed profile profiles/aggregation_in_step.json (from {in_step.get('source')}; rocprofv3 --kernel-trace of the training step)"},
            "cold_device_copy": {"GBps": copy_gbs, "frac_of_peak": copy_gbs / HBM_PEAK_GBS,
                                 "what": "torch copy_ of one message buffer into another under the same rotation (read + write bytes)",
                                 "aggregation_vs_copy": gbs(ms) / copy_gbs}}
    # edge-MLP forward through a GMP at L0: flops of the three D x D Linears per edge row
    if bf:
        return roof, None
    gmp = eng.GMP(D, 3, wl["cfg"]["pos_dim"]).cuda()
    x = torch.randn(batch, n0, D, device="cuda")
    pos = wl["node_in"][..., wl["cfg"]["out_dim"]: wl["cfg"]["out_dim"] + wl["cfg"]["pos_dim"]].contiguous()
    with torch.no_grad():
        msf = time_kernel(lambda: gmp(x, g0, pos, plan=plan), iters=10, warm=2)
    p = wl["cfg"]["pos_dim"]
    flops = 2 * batch * (e0 * (3 * D * D) + n0 * (2 * D * D + 2 * D * D + 3 * D * D))  # as executed (layer 0 hoisted to nodes)
    mf = {"kernel": "GMP forward at L0 (prepack + proj + k_chain_fwd edge/node + aggregation)", "bound": "mfma",
          "achieved": flops / (msf * 1e-3) / 1e12, "peak": MFMA_SPLIT_PEAK_TF, "unit": "TFLOP/s (fp32 flops)",
          "frac": flops / (msf * 1e-3) / 1e12 / MFMA_SPLIT_PEAK_TF, "flops": flops, "avg_us": msf * 1e3,
          "peak_note": "dense f16 MFMA peak / 3 (three fp16 partial products per fp32 multiply-add); "
                       f"for reference the f32-input MFMA peak is {MFMA_F32_PEAK_TF} TFLOP/s"}
    return roof, mf


def optimizer_step_time(dp, iters=20):
    """Reported next to the metric, not part of it: global-norm clip + AdamW over the flat
    parameter / gradient buffers (bsms_adamw_step), average of `iters` steps in microseconds."""
    import bsms_gnn_amd as eng
    opt = eng.FusedAdamW(dp.grads, lr=1e-4, weight_decay=1e-4, max_grad_norm=1.0)
    ms = time_kernel(lambda: opt.step(1e-9), iters=iters, warm=3)     # lr ~ 0: the parameters stay put
    return {"avg_us": ms * 1e3, "what": "clip_grad_norm_(1.0) + AdamW over all trainable parameters, fused (2 launches)"}


def rollout_rate(sim, wl, steps=200):
    """Secondary figure: forward-only autoregressive rollout, batch 1 like the reference
    (rollout.py:48), inference mode; eager launches vs replay of one captured HIP graph per step (capture excluded)."""
    from bsms_gnn_amd.rollout import _Stepper
    c = wl["cfg"]["out_dim"]
    ic, mask = wl["node_in"][:1].contiguous(), wl["mask"][:1].contiguous()
    g1, i1 = [g[:1] for g in wl["m_gs"]], [i[:1] for i in wl["m_ids"]]
    out = {"batch": 1, "steps": steps, "unit": "rollout steps/s (forward only)"}
    with torch.no_grad():
        for name, use_graph in (("eager", False), ("hip_graph", True)):
            st = _Stepper(sim, ic, mask, g1, i1, c, use_graph)
            for _ in range(5):
                st.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                st.step()
            torch.cuda.synchronize()
            out[name] = steps / (time.perf_counter() - t0)
    return out


def timed_steps(step, warmup, steps):
    """steps/s of `step()` over a short region: `warmup` untimed steps, then `steps` steps between two synchronisations."""
    for _ in range(warmup):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return steps / dt, dt / steps * 1e3, float(loss.detach())


def other_lines(sim_airfoil, data_airfoil, loss_f32, warmup=10, steps=30):
    """The other BASELINE configurations on the driver's line (N = 1, default run only): each its own short timed region
    (10 warm-up + 30 steps, wall clock between two synchronisations), same step definition as `value`.
      airfoil_bf16 / airfoil_bf16_nodes   BASELINE configs[2]: the headline workload in the BSMS_BF16 / BSMS_BF16_NODES precision
      surface_b2_bf16                      configs[4] per GPU: 16384 nodes, 6 levels, D=256, pos_dim 3, batch 2
      cylinder_b8 / cylinder_b8_blockdiag  configs[1]: dense batch of one mesh / 8 different meshes in one block-diagonal batch
      cylinder_b8_fresh_batches_*          configs[1] as the reference trains it: a NEW combination of 8 meshes every step
    `loss_vs_f32` = |loss - fp32 loss| / fp32 loss of the SAME engine, model and batch; `loss_vs_oracle` (cylinder, dense) =
    against the CPU oracle's forward on the same seed."""
    import gc
    import bsms_gnn_amd as eng
    out = {"protocol": f"{warmup} warm-up + {steps} timed steps per line, wall clock between two synchronisations; N = 1"}

    def run(sim, data, consistent, dtype):
        sim.process.precision = dtype
        dp = eng.DataParallel(sim)
        v, ms, loss = timed_steps(lambda: dp.step_loss_backward(data, consistent), warmup, steps)
        return {"value": v, "unit": "steps/s", "ms_per_step": ms, "steps": steps, "dtype": dtype, "loss": loss}

    for dtype in ("bf16", "bf16_nodes"):
        r = run(sim_airfoil, data_airfoil, True, dtype)
        r["loss_vs_f32"] = abs(r["loss"] - loss_f32) / abs(loss_f32)
        out[f"airfoil_{dtype}"] = r
    sim_airfoil.process.precision = "f32"
    gc.collect(); torch.cuda.empty_cache()

    def fresh(kind, batch, blockdiag=False):
        wl = build_workload(kind, batch, "cuda")
        torch.manual_seed(0)
        sim = eng.BSMS_Simulator(make_cfg(wl["cfg"])).cuda()
        if blockdiag:
            data, consistent = build_blockdiag_workload(kind, batch, "cuda")["data"], False
        else:
            data, consistent = data_tuple(wl), True
        sim(data, consistent, True)
        return wl, sim, data, consistent

    wl, sim, data, consistent = fresh("surface", 2)
    f32 = run(sim, data, consistent, "f32")
    r = run(sim, data, consistent, "bf16")
    r["loss_vs_f32"] = abs(r["loss"] - f32["loss"]) / abs(f32["loss"])
    out["surface_b2_bf16"], out["surface_b2_f32"] = r, f32
    del sim, data, wl
    gc.collect(); torch.cuda.empty_cache()

    wl, sim, data, consistent = fresh("cylinder", 8)
    r = run(sim, data, consistent, "f32")
    try:   # the CPU oracle's forward on the same seed and batch (no backward: ~1 s)
        from oracle import bsms_oracle as ro
        wc = build_workload("cylinder", 8, "cpu")
        torch.manual_seed(0)
        ref = ro.BSMS_Simulator(make_cfg(wc["cfg"]))
        ref(data_tuple(wc), True, True)
        with torch.no_grad():
            lo = float(ro.masked_rmse(ref(data_tuple(wc), True, False), wc["target"], wc["mask"]))
        r["loss_vs_oracle"] = abs(r["loss"] - lo) / abs(lo)
    except Exception as e:  # noqa: BLE001  (the line is still worth printing)
        r["loss_vs_oracle"] = f"not computed: {e}"
    out["cylinder_b8"] = r
    del sim, data
    gc.collect(); torch.cuda.empty_cache()
    wl, sim, data, consistent = fresh("cylinder", 8, blockdiag=True)
    out["cylinder_b8_blockdiag"] = run(sim, data, consistent, "f32")
    # Round 6: the reference's ACTUAL cylinder path -- a shuffled loader hands the trainer a NEW combination of meshes every step
    # (consistent_mesh: false, datasets/base.py:319-351), so nothing mesh-dependent of the BATCH is cached.  16 meshes, a random
    # 8 of them per step; (a) host collate + upload + plans from edge lists (rounds 3-5), (b) graph.MeshBank: per-mesh plans and edge
    # weights resident in HBM, the batch assembled by bsms_plan_concat.  Timed region = collate + step, no synchronisation inside.
    threads = torch.get_num_threads()
    try:
        # host-side tensor work (collate, hashing) with PyTorch's default of one intra-op thread per LOGICAL cpu gets a process on a small
        # cgroup quota throttled for whole scheduler periods (measured: 56 ms per step instead of 4; profiles/fresh_mesh.py does the same)
        torch.set_num_threads(max(1, min(usable_cpus() // 2, 8)))
        meshes = build_blockdiag_workload("cylinder", 16, "cpu")["samples"]
        sim.process.precision = "f32"
        dp = eng.DataParallel(sim)
        gen = torch.Generator().manual_seed(1)
        pick = lambda: [meshes[i] for i in torch.randperm(len(meshes), generator=gen)[:8].tolist()]
        bank = eng.MeshBank(sim.process, "cuda")
        host_batch = lambda: [d.to("cuda", intern=True) for d in eng.collate_variable_meshes(pick())]
        for name, make in (("host_collate", host_batch), ("mesh_bank", lambda: bank.collate(pick()))):
            v, ms, loss = timed_steps(lambda: dp.step_loss_backward(make(), False), 2 * warmup, steps)
            out[f"cylinder_b8_fresh_batches_{name}"] = {"value": v, "unit": "steps/s", "ms_per_step": ms, "steps": steps, "dtype": "f32", "loss": loss,
                                                        "what": "every step a new random 8 of 16 different cylinder meshes; timed: collate + upload + plans + step"}
    except Exception as e:  # noqa: BLE001
        out["cylinder_b8_fresh_batches"] = f"not measured: {e}"
    finally:
        torch.set_num_threads(threads)
    return out


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` from a plain shell (no torchrun): start N ranks of this script, one per GPU, wired up
    through the same environment variables torch.distributed.run sets (the reference never got this far: it wraps
    nn.DataParallel and then pins one GPU, trainer/trainer.py:15-18, train.py:16).  Backend nccl (= RCCL over xGMI) when
    the box has N GPUs; with fewer, the ranks share devices (rank % device_count) over gloo -- a FUNCTIONAL run of the
    N > 1 path, labelled as such (`rccl_ranks`: 0) because two RCCL ranks cannot share a device."""
    import subprocess
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (the BSMS engine has no CPU path)")
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), BSMS_BENCH_LAUNCHER="self")
    if ndev < n:
        env["BSMS_DIST_BACKEND"] = "gloo"
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cpus() // n)))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        if ndev < n:
            e["BSMS_FORCE_DEVICE"] = str(r % ndev)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=e))
    rc = 0
    try:
        while procs:
            for pr in list(procs):
                code = pr.poll()
                if code is None:
                    continue
                procs.remove(pr)
                if code != 0:                     # one rank failed: the others would wait for it in a collective forever
                    rc = rc or code
                    for other in procs:
                        other.terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            pr.kill()
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="airfoil", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=8, help="batch per GPU")
    ap.add_argument("--layout", default="dense", choices=["dense", "blockdiag"],
                    help="dense: consistent mesh [B,N,.]; blockdiag: B different meshes as one block-diagonal graph")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "bf16_nodes"],
                    help="f32: the reference's arithmetic (the headline line).  bf16: BSMS_BF16 precision of the U-Net "
                         "(BASELINE configs[2]/[4]; a SEPARATE line).  bf16_nodes: BSMS_BF16_NODES (node MLP in bf16 too)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-steps", type=int, default=0,
                    help="CPU baseline by SURVEY.md 8(d)'s protocol: 2 warm-up steps + best of K timed ones (default 0: a ~30 s sample)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--roofline-only", action="store_true", help="only the kernel micro-loops (for rocprofv3 --pmc passes)")
    ap.add_argument("--no-other-lines", action="store_true", help="skip the short timed regions of the other BASELINE configurations")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the BSMS engine has no CPU path)")
    backend = os.environ.get("BSMS_DIST_BACKEND", "nccl")       # "gloo": functional check of the N > 1 path on one GPU
    if os.environ.get("BSMS_FORCE_DEVICE") is not None:
        local = int(os.environ["BSMS_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.set_num_threads(max(1, min(8, usable_cpus() // world)))   # eight ranks share one host (and its cgroup quota)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start bench.py either plainly (it launches its own ranks) "
                         f"or under torch.distributed.run with --nproc-per-node {args.gpus}")

    import bsms_gnn_amd as eng
    wl = build_workload(args.workload, args.batch, "cuda", seed=rank)   # each rank its own samples of the shared mesh
    if args.roofline_only:
        roof, mf = roofline_objects(wl, args.batch, args.dtype)
        print(json.dumps({"roofline": roof, "roofline_mfma": mf}))
        return
    torch.manual_seed(0)
    sim = eng.BSMS_Simulator(make_cfg(wl["cfg"])).cuda()
    sim.process.precision = args.dtype
    consistent = args.layout == "dense"
    if consistent:
        data = data_tuple(wl)
    else:
        bd = build_blockdiag_workload(args.workload, args.batch, "cuda")
        data, wl["levels"] = bd["data"], bd["levels"]
    sim(data, consistent, True)                                           # one normaliser accumulation
    dp = eng.DataParallel(sim)
    dp.sync_normalizers()

    def step():
        return dp.step_loss_backward(data, consistent)

    if world > 1:          # FusedStep measures its two all-reduce forms during its first four data-parallel steps (synchronised): before the warm-up
        for _ in range(4):
            step()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    h0 = time.perf_counter()                 # host cost of a step: enqueue 3 steps into an EMPTY queue, no synchronisation
    for _ in range(3):
        step()
    host_ms = (time.perf_counter() - h0) / 3 * 1e3
    # ---- the timed region (contract): barrier + synchronize, EXACTLY `steps` steps, synchronize + barrier; max over ranks.
    # Inside it every step is also bracketed by HIP events on the launching stream (no synchronisation, a few hundred ns of
    # host time each): their MEDIAN is the per-step figure SURVEY.md section 8(d) asks for, reported next to the wall clock.
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        loss = step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_step = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]) if args.steps else np.zeros(1)
    allreduce_us = None
    if world > 1:
        t = torch.tensor([elapsed, float(np.median(per_step)), host_ms], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, med_ms, host_ms = float(t[0]), float(t[1]), float(t[2])   # host enqueue: the slowest rank's
        # the gradient message on its own: 20 back-to-back all-reduces of the flat gradient buffer inside one HIP event pair
        for _ in range(3):
            dist.all_reduce(dp.grads.flat)
        torch.cuda.synchronize()
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        for _ in range(20):
            dist.all_reduce(dp.grads.flat)
        eb.record()
        torch.cuda.synchronize()
        t = torch.tensor([ea.elapsed_time(eb) / 20 * 1e3], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        allreduce_us = float(t[0])
    else:
        med_ms = float(np.median(per_step))

    if rank == 0:
        n_params = sum(p.numel() for p in sim.parameters() if p.requires_grad)
        line = {
            "metric": f"rollout steps/sec (1-step fwd+bwd) on {args.workload} mesh, batch={args.batch}",
            "value": world * args.steps / elapsed, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "step_ms_hipevent": {"median": med_ms, "min": float(per_step.min()), "p90": float(np.percentile(per_step, 90)),
                                 "steps_per_s_from_median": world * 1e3 / med_ms if med_ms > 0 else None,
                                 "what": "every timed step bracketed by HIP events on the launching stream (rank 0 min / p90; median = max over ranks)"},
            "arithmetic": ("fp32 in/out/accumulate; matrix products as three partial products of two-way fp16 splits of power-of-two-scaled fp32 operands on v_mfma_f32_16x16x32_f16 (error <= f32 MFMA and <= fp32 FMA chain, profiles/census/f16split.hip)"
                           if args.dtype == "f32" else
                           "BSMS_BF16_NODES: BSMS_BF16 plus the node MLP -- its Linears multiply bf16 operands ([x, aggr] and hidden activations "
                           "rounded as they enter, fp32 accumulation), hidden activations and layer gradients gN[1..H] stored as bf16; block "
                           "input / output rows, residuals, projections, LayerNorm, aggregation sums, encoder / decoder, loss fp32"
                           if args.dtype == "bf16_nodes" else
                           "BSMS_BF16: edge activations / messages / edge layer gradients stored as bf16, edge-MLP products bf16 x bf16 "
                           "with fp32 accumulation; node level, LayerNorm, aggregation sums, encoder / decoder, loss, weight-gradient "
                           "accumulators fp32 (no reference parity target: the reference is fp32 only)"),
            "config": {"workload": f"{args.workload}-like Delaunay mesh, {wl['cfg']['nodes']} nodes, "
                                   f"{wl['cfg']['levels']} bi-stride levels, D={wl['cfg']['latent']}, hidden_layer=3, "
                                   f"batch {args.batch} per GPU (global {args.batch * world}), "
                                   + ("consistent mesh" if consistent else "block-diagonal batch of different meshes"),
                       "levels_N_E": wl["levels"], "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "trainable_params": n_params, "loss": float(loss.detach())},
            "host_enqueue_ms_per_step": host_ms,
        }
        if world > 1:
            probe = getattr(dp.fused, "_ov_probe", None) if dp.fused is not None else None
            line["distributed"] = {"backend": backend, "rccl_ranks": world if backend == "nccl" else 0,
                                   "gradient_allreduce": ("per bucket under the backward" if probe and probe.get("use") else "one message after the backward"),
                                   "allreduce_self_check_ms": None if not probe else probe.get("measured_ms"),
                                   "gradient_allreduce_us": allreduce_us, "gradient_bytes": int(dp.grads.flat.numel()) * 4,
                                   "host_enqueue_ms_per_step_max_over_ranks": host_ms,
                                   "launcher": os.environ.get("BSMS_BENCH_LAUNCHER", "torch.distributed.run"),
                                   "devices": torch.cuda.device_count(),
                                   "note": None if backend == "nccl" else
                                   "ranks share devices over gloo: a functional run of the N > 1 path, not a scaling measurement"}
        if world == 1 and not args.no_roofline and consistent:
            line["roofline"], line["roofline_mfma"] = roofline_objects(wl, args.batch, args.dtype)
            line["rollout"] = rollout_rate(sim, wl)
        if world == 1:
            line["optimizer_step"] = optimizer_step_time(dp)
        if world == 1 and consistent and args.workload == "airfoil" and args.dtype == "f32" and args.batch == 8 and not args.no_other_lines:
            line["other_lines"] = other_lines(sim, data, line["config"]["loss"])
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cb = cpu_baseline(args.workload, args.batch, steps=args.cpu_baseline_steps)
            if consistent:   # same seed, same workload: the oracle's loss IS the expected GPU loss (parity at bench size)
                rel = abs(cb["loss"] - line["config"]["loss"]) / abs(cb["loss"])
                line["config"]["loss_vs_cpu_oracle_rel"] = rel
                assert rel <= (1e-5 if args.dtype == "f32" else 1e-2), f"GPU loss {line['config']['loss']} != CPU oracle loss {cb['loss']} (rel {rel:.2e})"
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
