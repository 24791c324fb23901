"""Full-size GPU parity: every BASELINE.json configuration, one whole training step (BSMS_Simulator forward +
masked RMSE + backward, reference: src/ops/BSMS.py:39-104, src/models/model.py:127-164, src/trainer/trainer.py:96-97)
on the HIP engine against the CPU oracle with identical seeded weights and inputs.

  airfoil   5233 nodes, 5 levels, D=128, B=8, consistent mesh           (configs[2]/[3]; the bench workload)
  cylinder  1885 nodes, 4 levels, D=128, B=8, consistent mesh           (configs[1])
  cylinder  the same as 8 DIFFERENT meshes in one block-diagonal batch  (configs[1], the reference's cylinder layout)
  surface   16384 nodes, 6 levels, D=256, pos_dim=3, B=2                (configs[4], per-GPU share of B=16 over 8 GPUs)

Tolerances (BASELINE.json north_star: 1e-5 relative fp32)
  * prediction: element-wise |gpu - cpu| <= 1e-5 |cpu| + 1e-5 * rms(cpu)   and   max|gpu - cpu| <= 1e-5 max|cpu|
  * loss: |gpu - cpu| <= 1e-5 |cpu|
  * gradients: THREE-WAY against an fp64 run of the same oracle ("exact").  At these sizes the reference's OWN fp32
    arithmetic is 1e-5 .. 1e-4 away from the exact gradient in most parameter tensors (measured: airfoil worst 9.2e-5,
    73 of 192 tensors above 1e-5) -- a step has ~5e8 ReLU inputs, a few dozen of which lie within fp32 round-off of 0,
    and which side of the kink they land on differs between ANY two fp32 summation orders -- so "GPU within 1e-5 of
    CPU fp32" is not a property even two runs of the reference on different BLAS builds have.  What is required:
        e_gpu(p) = max|g_gpu - g_64| / max|g_64| ,  e_cpu(p) likewise for the fp32 oracle
        e_cpu = the fp32 oracle run TWICE, on all host threads and on one thread (two summation orders of the same arithmetic);
        every statistic below takes the LARGER of the two runs' values (round 6)
        (a) worst tensor:    max_p e_gpu            <= max(1e-5, 1.5 * max_p e_cpu)
        (b) typical tensor:  median_p e_gpu         <= max(1e-5, 1.5 * median_p e_cpu)
        (c) tail:            90th percentile e_gpu  <= max(1e-5, 1.5 * 90th percentile e_cpu)
        (d) regression guard on the DIRECT distance: median_p max|g_gpu - g_cpu32| / max|g_cpu32| <= max(1e-5, 2.5 * median_p e_cpu)
            (triangle inequality: two fp32 paths that are each e away from the exact gradient are at most 2e apart)
    i.e. the engine's distance to the exact gradient has the same distribution over the parameter tensors as the
    reference's own fp32 path.  (Rounds 1-3 allowed 3x / 2x / 2x; the measured ratios are 1.0-1.4 on the BASELINE
    configurations.  The depth-7 strip at B=4 -- few rows per coarse level, so few independent kink events -- measured a
    median ratio of 1.63 in round 4 and 0.22 in round 5 against a single oracle run, whose own distance moves with the host's
    thread count; rounds 4-5 therefore allowed 2x for (b) there.  Since round 6 the yardstick is two oracle runs and every
    configuration uses 1.5; the measured ratios are printed and recorded in profiles/r06_tests.txt.)  A per-tensor ratio is not meaningful (which tensors a near-zero ReLU input lands in is
    random for both implementations), and the factors allow for the fact that ONE flipped mask perturbs the gradient
    of every layer upstream of it, so the per-tensor errors of a run are strongly correlated (few independent events).
    Measured on MI355X (gpu | cpu32, worst / median): airfoil B=8 9.4e-5 / 1.0e-5 | 9.3e-5 / 8.1e-6; cylinder B=8
    2.2e-4 / 4.5e-5 | 2.2e-4 / 4.5e-5; cylinder block-diagonal 3.1e-4 / 6.2e-5 | 3.1e-4 / 6.1e-5; surface B=2
    2.6e-4 / 6.0e-5 | 6.0e-4 / 4.6e-5; depth-7 strip 4.7e-4 / 5.7e-5 | 3.4e-4 / 3.5e-5.  Loss equal to <= 8e-8, pred to <= 2.5e-7.
"""
import os
import time

import numpy as np
import pytest
import torch

from oracle import bsms_oracle as ro

pytestmark = pytest.mark.gpu


def _to(obj, fn):
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to(o, fn) for o in obj)
    return fn(obj)


def _f64(t):
    return t.double() if t.is_floating_point() else t


def _oracle_step(sim, data, consistent):
    sim.zero_grad(set_to_none=True)
    pred = sim(data, consistent, False)
    loss = ro.masked_rmse(pred, data[1], data[2])
    loss.backward()
    return pred.detach(), float(loss.detach()), {k: p.grad.clone() for k, p in sim.named_parameters() if p.grad is not None}


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300))


def run_config(eng, kind, batch, layout, mesh=None, cfg=None):
    from bench import build_blockdiag_workload, build_workload, data_tuple, make_cfg, usable_cpus
    torch.set_num_threads(max(1, min(32, usable_cpus())))
    if layout == "dense":
        wl = build_workload(kind, batch, "cpu", mesh=mesh, cfg=cfg)
        cpu_data, consistent = data_tuple(wl), True
        gpu_data = _to(cpu_data, lambda t: t.cuda())
    else:
        wl = build_blockdiag_workload(kind, batch, "cpu")
        lv = wl["data"]
        cpu_data = (lv[0].x.unsqueeze(0), lv[0].y.unsqueeze(0), lv[0].mask.unsqueeze(0), [d.edge_index for d in lv],
                    [d.face for d in lv[:-1]])
        consistent = False
        gpu_data = [d.to("cuda") for d in lv]
    cfg = make_cfg(wl["cfg"])
    torch.manual_seed(0)
    ref32 = ro.BSMS_Simulator(cfg)
    ref32(cpu_data, consistent, True)                      # one normaliser accumulation (fp64 statistics)
    ref64 = ro.BSMS_Simulator(cfg, dtype=torch.float64)
    ref64.load_state_dict(ref32.state_dict())
    ref64.double()
    mine = eng.BSMS_Simulator(cfg)
    mine.load_state_dict(ref32.state_dict())
    mine = mine.cuda()

    t0 = time.perf_counter()
    pred32, loss32, g32 = _oracle_step(ref32, cpu_data, consistent)
    t32 = time.perf_counter() - t0
    _, loss64, g64 = _oracle_step(ref64, _to(cpu_data, _f64), consistent)
    t64 = time.perf_counter() - t0 - t32
    # the SAME fp32 oracle on one thread: another summation order of the same arithmetic (round 6; the yardstick of `check`)
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        _, _, g32_one = _oracle_step(ref32, cpu_data, consistent)
    finally:
        torch.set_num_threads(nthreads)

    pred = mine(gpu_data, consistent, False)
    tar, mask = (gpu_data[1], gpu_data[2]) if consistent else (gpu_data[0].y.unsqueeze(0), gpu_data[0].mask.unsqueeze(0))
    loss = eng.masked_rmse(pred, tar, mask)
    loss.backward()
    torch.cuda.synchronize()
    gg = {k: p.grad.detach().cpu() for k, p in mine.named_parameters() if p.grad is not None}
    return dict(pred=pred.detach().cpu(), pred32=pred32, loss=float(loss.detach()), loss32=loss32, loss64=loss64,
                gg=gg, g32=g32, g32_one=g32_one, g64=g64, t32=t32, t64=t64, levels=wl["levels"])


def check(r, tag):
    F_WORST = F_MEDIAN = F_P90 = 1.5     # one set of factors for every configuration (round 6: no per-test overrides)
    F_DIRECT = 2.5
    # ---- forward: element-wise and max-norm, against the fp32 oracle
    a, b = r["pred"].double(), r["pred32"].double()
    scale, rms = float(b.abs().max()), float(b.pow(2).mean().sqrt())
    diff = (a - b).abs()
    assert float(diff.max()) <= 1e-5 * scale, (tag, "pred max-norm", float(diff.max()) / scale)
    bad = diff > 1e-5 * b.abs() + 1e-5 * rms
    assert not bool(bad.any()), (tag, "pred element-wise", int(bad.sum()))
    assert abs(r["loss"] - r["loss32"]) <= 1e-5 * abs(r["loss32"]), (tag, r["loss"], r["loss32"])
    # ---- gradients: three-way against fp64 (criterion in the module docstring)
    assert set(r["gg"]) == set(r["g32"]) == set(r["g64"]) == set(r["g32_one"]), "a parameter is missing its gradient"
    keys = sorted(r["g64"])
    e_gpu = np.array([_rel(r["gg"][k], r["g64"][k]) for k in keys])
    e_cpuN = np.array([_rel(r["g32"][k], r["g64"][k]) for k in keys])        # the oracle on all threads
    e_cpu1 = np.array([_rel(r["g32_one"][k], r["g64"][k]) for k in keys])    # the oracle on one thread
    direct = np.array([_rel(r["gg"][k], r["g32"][k]) for k in keys])
    worst = int(np.argmax(e_gpu))
    # Yardstick = the LARGER of the fp32 oracle's two distances to fp64, statistic by statistic: which near-zero ReLU inputs flip
    # is a property of the summation order, the oracle has (at least) these two, and the engine is a third
    y_worst = max(e_cpuN.max(), e_cpu1.max())
    y_median = max(np.median(e_cpuN), np.median(e_cpu1))
    y_p90 = max(np.percentile(e_cpuN, 90), np.percentile(e_cpu1, 90))
    p90g = np.percentile(e_gpu, 90)
    print(f"\n[{tag}] levels {r['levels']}\n  oracle fp32 {r['t32']:.1f} s, fp64 {r['t64']:.1f} s; loss gpu {r['loss']:.7f} "
          f"cpu32 {r['loss32']:.7f} f64 {r['loss64']:.7f}; pred max-norm {float(diff.max()) / scale:.2e}\n"
          f"  grads vs fp64 over {len(keys)} tensors: gpu worst {e_gpu.max():.2e} ({keys[worst]}) median {np.median(e_gpu):.2e} p90 {p90g:.2e} | "
          f"cpu32 all threads worst {e_cpuN.max():.2e} median {np.median(e_cpuN):.2e} p90 {np.percentile(e_cpuN, 90):.2e} | "
          f"cpu32 one thread worst {e_cpu1.max():.2e} median {np.median(e_cpu1):.2e} p90 {np.percentile(e_cpu1, 90):.2e} | "
          f"gpu-vs-cpu32 worst {direct.max():.2e} median {np.median(direct):.2e}\n"
          f"  ratios gpu / yardstick: worst {e_gpu.max() / y_worst:.2f} median {np.median(e_gpu) / y_median:.2f} p90 {p90g / y_p90:.2f} "
          f"direct-median {np.median(direct) / y_median:.2f}   (limits {F_WORST} / {F_MEDIAN} / {F_P90} / {F_DIRECT})")
    assert e_gpu.max() <= max(1e-5, F_WORST * y_worst), (tag, "worst tensor", keys[worst], e_gpu.max(), y_worst)
    assert np.median(e_gpu) <= max(1e-5, F_MEDIAN * y_median), (tag, "median", np.median(e_gpu), y_median)
    assert p90g <= max(1e-5, F_P90 * y_p90), (tag, "90th percentile", p90g, y_p90)
    assert np.median(direct) <= max(1e-5, F_DIRECT * y_median), (tag, "gpu-vs-cpu32 median", np.median(direct), y_median)


@pytest.fixture(scope="module")
def eng():
    import bsms_gnn_amd as eng
    return eng


def test_airfoil_b8_step_matches_oracle(eng):
    r = run_config(eng, "airfoil", 8, "dense")
    assert r["levels"][0] == (5233, 31354) and len(r["levels"]) == 6
    check(r, "airfoil B=8 L=5 D=128")


def test_cylinder_b8_dense_step_matches_oracle(eng):
    r = run_config(eng, "cylinder", 8, "dense")
    assert r["levels"][0] == (1885, 11264) and len(r["levels"]) == 5
    check(r, "cylinder B=8 L=4 D=128 dense")


def test_cylinder_b8_blockdiag_step_matches_oracle(eng):
    r = run_config(eng, "cylinder", 8, "blockdiag")
    assert r["levels"][0][0] == 8 * 1885 and len(r["levels"]) == 5
    check(r, "cylinder 8 different meshes, block-diagonal")


def test_surface_b2_step_matches_oracle(eng):
    r = run_config(eng, "surface", 2, "dense")
    assert r["levels"][0][0] == 16384 and len(r["levels"]) == 7 and r["levels"][-1][0] >= 2
    check(r, "surface B=2 L=6 D=256 p=3")


def test_airfoil_depth7_reference_default(eng):
    """configs/model/airfoil.yaml:4 `unet_depth: 7` (the reference's default) on an airfoil-sized mesh that supports
    it (bench.strip_mesh: 5232 nodes, 32 at level 7): one B=4 step matches the oracle, same rules."""
    from bench import strip_mesh
    w, mesh = strip_mesh(327, 16, 7)
    r = run_config(eng, "airfoil", 4, "dense", mesh=mesh, cfg=w)
    assert len(r["levels"]) == 8 and r["levels"][0][0] == 5232 and r["levels"][-1][0] >= 2
    # (rounds 4-5 allowed 2.0 for the median here: the ratio against ONE oracle run measured 1.63 on one box and 0.22 on another --
    # the oracle's own summation order moves with the host's thread count.  Round 6: the yardstick is the larger of two oracle
    # runs, one thread and all threads, and the factors are the same 1.5 as everywhere.)
    check(r, "airfoil-sized strip B=4 L=7 (reference default depth)")
