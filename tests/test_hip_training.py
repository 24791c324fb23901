"""GPU: fused optimizer step and the Trainer loop against torch.optim.AdamW / the CPU oracle."""
import copy
from types import SimpleNamespace

import pytest
import torch

from conftest import load_golden, rel_err
from oracle import bsms_oracle as ro

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import bsms_gnn_amd as eng
    return eng


def test_fused_adamw_matches_torch(eng):
    torch.manual_seed(0)
    shapes = [(128, 259), (128,), (128, 128), (3, 128), (3,)]
    cpu = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    gpu = [torch.nn.Parameter(p.detach().clone().cuda()) for p in cpu]
    ref = torch.optim.AdamW(cpu, lr=1e-3, weight_decay=1e-2)
    buckets = eng.GradBuckets(gpu)
    opt = eng.FusedAdamW(buckets, lr=1e-3, weight_decay=1e-2, max_grad_norm=1.0)
    for step in range(4):
        buckets.zero()
        grads = [torch.randn(s) * (3.0 if step % 2 else 0.01) for s in shapes]    # clipped and unclipped steps
        for p, q, g in zip(cpu, gpu, grads):
            p.grad = g.clone()
            off, n = buckets._slot[q]
            buckets.flat[off:off + n].copy_(g.reshape(-1).cuda())
        total = torch.nn.utils.clip_grad_norm_(cpu, 1.0)
        ref.step()
        opt.step()
        assert abs(float(opt.grad_norm) - float(total)) < 1e-5 * float(total)
        for p, q in zip(cpu, gpu):
            assert rel_err(q.detach().cpu(), p.detach()) < 2e-6, step
    assert all(q.data_ptr() >= opt.flat_p.data_ptr() for q in gpu)                # parameters live in the flat array


def test_lr_schedule_matches_reference_formula(eng):
    sch = eng.WarmupCosineDecay(1e-4, 10, 100)
    import numpy as np
    for epoch in range(0, 100, 7):
        want = epoch / 10 if epoch <= 10 else 0.5 * (1 + np.cos(np.pi * (epoch - 10) / 90))    # utils/basic.py:177-184
        assert abs(sch.factor(epoch) - want) < 1e-12
    assert sch.lr() == 0.0                                                          # first step sees epoch 0


def test_trainer_iterations_follow_cpu_reference_loop(eng, graphs, tmp_path):
    """Warm-up + 4 optimisation steps of Trainer.iter == the same loop written with the CPU oracle,
    torch clip_grad_norm_ and torch.optim.AdamW (trainer/trainer.py:134-156)."""
    z = load_golden("sim")
    es, ids = graphs.levels("del300")
    B = 2
    cfg = ro.make_cfg(2, 32, 3, 3, 2)
    model_cfg = SimpleNamespace(consistent_mesh=True, accumulation_steps=1)
    opt_cfg = SimpleNamespace(peak_lr=1e-3, weight_decay=1e-4, warmup_steps=2, decay_steps=20, gnorm_clip=1.0)
    torch.manual_seed(0)
    ref = ro.BSMS_Simulator(cfg)
    mine = eng.BSMS_Simulator(cfg)
    mine.load_state_dict(ref.state_dict())
    m_gs = [e.unsqueeze(0).repeat(B, 1, 1) for e in es]
    m_ids = [i.unsqueeze(0).repeat(B, 1) for i in ids]
    data = (z.t("node_in"), z.t("tar"), z.t("mask"), m_gs, m_ids)

    tr = eng.Trainer(mine, model_cfg, opt_cfg)
    opt = torch.optim.AdamW([p for p in ref.parameters() if p.requires_grad], lr=opt_cfg.peak_lr, weight_decay=opt_cfg.weight_decay)
    sch = eng.WarmupCosineDecay(opt_cfg.peak_lr, opt_cfg.warmup_steps, opt_cfg.decay_steps)
    losses_ref, losses = [], []
    for step in range(5):
        if step < model_cfg.accumulation_steps:
            ref(data, True, True)
        else:
            opt.zero_grad()
            loss = ro.masked_rmse(ref(data, True, False), data[1], data[2])
            loss.backward()
            torch.nn.utils.clip_grad_norm_(ref.parameters(), opt_cfg.gnorm_clip)
            for gparam in opt.param_groups:
                gparam["lr"] = sch.lr()
            opt.step()
            sch.step()
            losses_ref.append(float(loss))
        out = tr.iter(data)
        if out is not None:
            losses.append(float(out))
    assert len(losses) == 4 and tr.train_step == 5
    # the CPU batch goes through move_to_device every iteration (trainer/trainer.py:143): the mesh plans must have been
    # built during the first iteration only (index tensors interned by content, graph.intern_index)
    from bsms_gnn_amd.graph import LevelPlan
    built = LevelPlan.constructed
    fresh = (data[0].clone(), data[1].clone(), data[2].clone(), [g.clone() for g in m_gs], [i.clone() for i in m_ids])
    tr.get_loss(fresh)                                   # NEW host tensors, same mesh: still no plan construction
    assert LevelPlan.constructed == built, "a consistent mesh rebuilt its plans"
    for a, b in zip(losses, losses_ref):
        assert abs(a - b) < 2e-4 * abs(b), (losses, losses_ref)
    assert losses[-1] < losses[1]                                                   # it actually learns
    for (k, p), (_, q) in zip(ref.named_parameters(), mine.named_parameters()):
        if p.requires_grad:
            assert rel_err(q.detach().cpu(), p.detach()) < 2e-3, k
    # save / restore incl. optimizer state (the reference leaves the latter as a TODO, trainer.py:188-193)
    tr.save(str(tmp_path))
    again = eng.Trainer(eng.BSMS_Simulator(cfg), model_cfg, opt_cfg)
    again.restore(str(tmp_path), tr.train_step)
    assert again.train_step == tr.train_step and again.lr_scheduler.last_epoch == tr.lr_scheduler.last_epoch
    l1, l2 = tr.iter(data), again.iter(data)
    assert abs(float(l1) - float(l2)) < 1e-6 * abs(float(l1))


@pytest.mark.parametrize("consistent", [True, False])
def test_datapipe_to_trainer_end_to_end(eng, consistent):
    """Synthetic trajectories -> datapipe (packing, masks, noise, hierarchy cache) -> loader -> Trainer.iter,
    for the consistent-mesh and the variable-mesh (block-diagonal) layouts."""
    import numpy as np
    import bsms_gnn_amd.datapipe as dpipe
    from test_datapipe import cfg as data_cfg, synthetic_traj
    dcfg = data_cfg(consistent)
    trajs = [synthetic_traj(120, 4, 9)] * 2 if consistent else [synthetic_traj(100, 4, 1), synthetic_traj(140, 4, 2)]
    ds = dpipe.TrajectoryDataset(dcfg, trajs, dataset="airfoil" if consistent else "cylinder_flow", mode="train", seed=0)
    loader = dpipe.make_loader(ds, 2)
    model_cfg = SimpleNamespace(out_dim=3, latent_dim=32, hidden_layer=2, unet_depth=2, pos_dim=2,
                                consistent_mesh=consistent, accumulation_steps=1)
    opt_cfg = SimpleNamespace(peak_lr=1e-3, weight_decay=1e-4, warmup_steps=1, decay_steps=50, gnorm_clip=1.0)
    torch.manual_seed(0)
    tr = eng.Trainer(eng.BSMS_Simulator(model_cfg), model_cfg, opt_cfg)
    losses = []
    from bsms_gnn_amd.graph import LevelPlan
    built = []
    for it, batch in enumerate(loader):
        out = tr.iter(batch)                             # host batches, moved (and interned) by the Trainer itself
        built.append(LevelPlan.constructed)
        if out is not None:
            losses.append(float(out))
    assert len(losses) >= 2 and all(np.isfinite(losses))
    if consistent:
        assert built[-1] == built[1], "plans of a consistent mesh are built once (warm-up builds none, step 1 builds them)"
    assert float(tr.optimizer.grad_norm) > 0
    # the same stream of batches behind the prefetch thread (uploads, plans and edge weights one batch ahead): same numbers
    ds2 = dpipe.TrajectoryDataset(dcfg, trajs, dataset="airfoil" if consistent else "cylinder_flow", mode="train", seed=0)
    torch.manual_seed(0)
    tr2 = eng.Trainer(eng.BSMS_Simulator(model_cfg), model_cfg, opt_cfg)
    losses2 = [float(out) for out in (tr2.iter(b) for b in eng.DevicePrefetcher(dpipe.make_loader(ds2, 2), tr2)) if out is not None]
    assert losses2 == losses
    with pytest.raises(ZeroDivisionError):                # an error in the loader surfaces in the training thread
        for _ in eng.DevicePrefetcher((1 // 0 for _ in range(1)), tr2):
            pass


def _sim_batch(eng, graphs, B=2):
    z = load_golden("sim")
    es, ids = graphs.levels("del300")
    m_gs = [e.unsqueeze(0).repeat(B, 1, 1).cuda() for e in es]
    m_ids = [i.unsqueeze(0).repeat(B, 1).cuda() for i in ids]
    return (z.t("node_in").cuda(), z.t("tar").cuda(), z.t("mask").cuda(), m_gs, m_ids)


def test_fused_step_equals_autograd_step(eng, graphs):
    """step.FusedStep (direct C-ABI calls, fused normaliser / loss kernels, gradients written into the flat buffer) against
    the autograd mirror of the reference on the same model and batch: loss, prediction and every parameter gradient.
    The U-Net / MLP kernels are the same; only the glue arithmetic is re-associated, hence 2e-6 instead of bit equality.
    The HIP-graph replay of the fused step must be bit-identical to its eager form."""
    import os
    cfg = ro.make_cfg(2, 32, 3, 3, 2)
    data = _sim_batch(eng, graphs)
    torch.manual_seed(3)
    sim = eng.BSMS_Simulator(cfg).cuda()
    sim(data, True, True)
    # autograd path
    sim.zero_grad(set_to_none=True)
    pred = sim(data, True, False)
    loss = eng.masked_rmse(pred, data[1], data[2])
    loss.backward()
    want = {k: p.grad.clone() for k, p in sim.named_parameters() if p.grad is not None}
    sim.zero_grad(set_to_none=True)
    # fused path
    grads = eng.GradBuckets(list(sim.parameters()))
    step = eng.FusedStep(sim, grads)
    got_loss = step(data, True)
    assert abs(float(got_loss) - float(loss)) < 1e-6 * abs(float(loss))
    assert rel_err(step.prediction().cpu(), pred.detach().cpu()) < 1e-6
    assert set(k for k, p in sim.named_parameters() if p.grad is not None) == set(want)
    for k, p in sim.named_parameters():
        if p.requires_grad:
            assert p.grad.data_ptr() >= grads.flat.data_ptr() and rel_err(p.grad.cpu(), want[k].cpu()) < 2e-6, k
    eager = grads.flat.clone()
    # a second batch shape / values through the same object, then the graph variant
    gstep = eng.FusedStep(sim, grads, use_graph=True)
    l1 = float(gstep(data, True))
    assert torch.equal(grads.flat, eager) and abs(l1 - float(got_loss)) == 0.0
    data2 = (data[0] * 1.01, data[1], data[2], data[3], data[4])
    l2 = float(gstep(data2, True))                        # replay with new inputs copied into the static buffers
    ref2 = float(step(data2, True))
    assert l2 == ref2 and l2 != l1


def test_fused_step_over_batches_of_changing_size(eng, graphs):
    """Variable meshes: every batch has its own node / edge counts.  The step's buffers are grow-only views (step._Arena);
    a step on a batch must not depend on which batches came before it (smaller, larger, the same again)."""
    cfg = ro.make_cfg(2, 32, 3, 2, 2)
    torch.manual_seed(11)
    sim = eng.BSMS_Simulator(cfg).cuda()

    def sample(nm, seed):
        es, ids = graphs.levels(nm)
        n = graphs.np(f"{nm}/pos").shape[0]
        gen = torch.Generator().manual_seed(seed)
        pos = torch.tensor(graphs.np(f"{nm}/pos")[:, :2], dtype=torch.float32)
        state = torch.randn(n, 2, generator=gen)
        x, y = torch.cat([state, pos, torch.zeros(n, 1)], -1), state + 0.1 * torch.randn(n, 2, generator=gen)
        sizes = [n] + [i.numel() for i in ids[:2]]
        return [eng.LevelData(es[l], sizes[l], face=ids[l] if l < 2 else None, x=x if l == 0 else None,
                              y=y if l == 0 else None, mask=torch.ones(n, 1) if l == 0 else None) for l in range(3)]

    small, big = sample("del64", 0), sample("del300", 1)
    batches = [eng.collate_variable_meshes(b) for b in ([small], [big, small], [big], [small, big, big], [small])]
    to_dev = lambda b: [d.to("cuda", intern=True) for d in b]
    sim(to_dev(batches[1]), False, True)                                   # normaliser statistics
    want = []
    for b in batches:                                                      # reference: a fresh step object per batch
        grads = eng.GradBuckets(list(sim.parameters()))
        want.append((float(eng.FusedStep(sim, grads)(to_dev(b), False)), grads.flat.clone()))
    grads = eng.GradBuckets(list(sim.parameters()))
    step = eng.FusedStep(sim, grads)
    for b, (loss, flat) in zip(batches, want):
        assert float(step(to_dev(b), False)) == loss and torch.equal(grads.flat, flat)
    caps = {k: t.numel() for k, t in step._arena._t.items()}
    step(to_dev(batches[0]), False)                                        # a small batch after the largest: nothing is re-allocated
    assert caps == {k: t.numel() for k, t in step._arena._t.items()}


def test_data_parallel_uses_the_fused_step(eng, graphs):
    """DataParallel.step_loss_backward == the autograd step (BSMS_FUSED_STEP=0 path), and the Trainer loop on top of it is
    covered by test_trainer_iterations_follow_cpu_reference_loop."""
    cfg = ro.make_cfg(2, 32, 3, 3, 2)
    data = _sim_batch(eng, graphs)
    torch.manual_seed(5)
    sim = eng.BSMS_Simulator(cfg).cuda()
    sim(data, True, True)
    dp = eng.DataParallel(sim)
    assert dp.fused is not None
    l_f = float(dp.step_loss_backward(data, True))
    g_f = dp.grads.flat.clone()
    dp.fused = None
    l_a = float(dp.step_loss_backward(data, True))
    assert abs(l_f - l_a) < 1e-6 * abs(l_a) and rel_err(g_f.cpu(), dp.grads.flat.cpu()) < 2e-6


def test_fused_step_rejects_misshaped_batches_and_returns_fresh_losses(eng, graphs):
    """FusedStep hands raw pointers with sizes from cfg to the C ABI: a batch that does not match cfg must raise before
    any launch (the autograd path would fail with a PyTorch shape error).  Returned losses are copies, not views of the
    static buffer the next step overwrites."""
    cfg = ro.make_cfg(2, 32, 3, 3, 2)
    data = _sim_batch(eng, graphs)
    torch.manual_seed(4)
    sim = eng.BSMS_Simulator(cfg).cuda()
    sim(data, True, True)
    step = eng.FusedStep(sim, eng.GradBuckets(list(sim.parameters())))
    node_in, tar, mask, m_gs, m_ids = data
    bad = [
        (node_in[..., :-1], tar, mask, m_gs, m_ids),                      # a feature column short
        (node_in, tar[..., :1], mask, m_gs, m_ids),                       # wrong out_dim
        (node_in, tar[:, :-1], mask, m_gs, m_ids),                        # a node short
        (node_in, tar, mask[:, :-1], m_gs, m_ids),                        # mask of another size
        (node_in, tar, mask, m_gs[:-1], m_ids),                           # a level short
        (node_in, tar.cpu(), mask, m_gs, m_ids),                          # mixed devices
    ]
    for d in bad:
        with pytest.raises(RuntimeError):
            step(d, True)
    l1 = step(data, True)
    v1 = float(l1)
    l2 = step((node_in * 1.5, tar, mask, m_gs, m_ids), True)
    assert float(l1) == v1 and float(l2) != v1                            # l1 survived the second step


def test_inference_session_sees_raw_pointer_updates(eng, graphs):
    """ops.InferenceSession reuses weight packs between forward-only calls; an optimizer step through the C ABI does not
    bump the tensors' version counters, so the session keys on the engine's parameter epoch as well."""
    from types import SimpleNamespace
    from bsms_gnn_amd.ops import InferenceSession
    cfg = ro.make_cfg(2, 32, 3, 3, 2)
    data = _sim_batch(eng, graphs)
    torch.manual_seed(6)
    tr = eng.Trainer(eng.BSMS_Simulator(cfg), SimpleNamespace(consistent_mesh=True, accumulation_steps=1),
                     SimpleNamespace(peak_lr=1e-2, weight_decay=1e-4, warmup_steps=1, decay_steps=20, gnorm_clip=1.0))
    tr.iter(data)
    node_in, tar, mask, m_gs, m_ids = data
    gs, ids = [g[0] for g in m_gs], [i[0] for i in m_ids]
    sess = InferenceSession(static_pos=True)
    with torch.no_grad():
        a = tr.model._infer(ids, gs, node_in, mask, session=sess).clone()
    tr.iter(data); tr.iter(data)                                          # lr > 0 from the second optimizer step on
    with torch.no_grad():
        b = tr.model._infer(ids, gs, node_in, mask, session=sess).clone()
        c = tr.model._infer(ids, gs, node_in, mask).clone()               # no session: packs rebuilt
    assert torch.equal(b, c) and not torch.equal(a, b)
