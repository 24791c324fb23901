"""Training harness around the hot path, mirroring the reference's `Trainer` (src/trainer/trainer.py) and
LR schedule (src/utils/basic.py:168-184): masked-RMSE loss, backward, global-norm clip, AdamW, warm-up +
cosine decay, save / restore.  The optimizer step is ONE fused HIP launch pair over the flat parameter and
gradient buffers (bsms_adamw_step) instead of ~180 small tensor updates; under data parallelism the gradients
are all-reduced (dp.py) before it, so every rank applies the identical update."""
import math
import os

import torch

from . import _abi
from .dp import DataParallel
from .ops import bump_param_epoch


class WarmupCosineDecay:
    """utils/basic.py:168-184: factor = epoch/warmup up to `warmup`, then 0.5 (1 + cos(pi * progress)).
    Like torch's _LRScheduler the first optimizer step sees epoch 0 (factor 0)."""

    def __init__(self, base_lr, warmup, max_iters):
        self.base_lr, self.warmup, self.max_iters, self.last_epoch = base_lr, warmup, max_iters, 0

    def factor(self, epoch=None):
        epoch = self.last_epoch if epoch is None else epoch
        if epoch <= self.warmup:
            return epoch * 1.0 / self.warmup
        return 0.5 * (1 + math.cos(math.pi * (epoch - self.warmup) / (self.max_iters - self.warmup)))

    def lr(self):
        return self.base_lr * self.factor()

    def step(self):
        self.last_epoch += 1


class FusedAdamW:
    """AdamW over the flat buffers of a GradBuckets (dp.py).  Parameters are re-pointed into one flat fp32
    array with the gradient buffer's layout, so clip + update are two kernel launches for the whole model."""

    def __init__(self, grads, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=0.0):
        self.grads, self.lr, self.betas, self.eps, self.wd, self.max_norm = grads, lr, betas, eps, weight_decay, max_grad_norm
        flat = torch.empty_like(grads.flat)
        for p in grads.params:
            off, n = grads._slot[p]
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view_as(p)
        self.flat_p = flat
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.step_count = 0
        self.grad_norm = torch.zeros(1, device=flat.device, dtype=torch.float32)
        self._work = torch.empty(max(int(_abi.lib().bsms_adamw_work_bytes()), 4), dtype=torch.uint8, device=flat.device)

    def step(self, lr=None):
        self.step_count += 1
        b1, b2 = self.betas
        _abi.check(_abi.lib().bsms_adamw_step(
            self.flat_p.data_ptr(), self.grads.flat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
            self.flat_p.numel(), float(self.lr if lr is None else lr), b1, b2, self.eps, self.wd, self.step_count,
            float(self.max_norm), self.grad_norm.data_ptr(), self._work.data_ptr(), torch.cuda.current_stream().cuda_stream),
            "bsms_adamw_step")
        bump_param_epoch()           # raw-pointer update: invalidates weight packs cached by ops.InferenceSession

    def state_dict(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self.step_count}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count = int(sd["step"])


def usable_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _check_host_threads():
    """PyTorch defaults to one intra-op thread per LOGICAL cpu; under a cgroup CPU quota (containers) the bursts of a few
    CPU tensor copies per step then exhaust the quota and the scheduler parks the whole process for the rest of the
    period -- 80 ms stalls on arbitrary lines of the training loop (profiles/fresh_mesh.py).  Warn once."""
    import warnings
    n = usable_cpus()
    if torch.get_num_threads() > 2 * n:
        warnings.warn(f"torch uses {torch.get_num_threads()} intra-op threads but this process may use ~{n} CPUs (cgroup quota / "
                      f"affinity): call torch.set_num_threads({max(1, min(n, 8))}) to avoid CFS throttling stalls in the training loop",
                      RuntimeWarning, stacklevel=3)


class DevicePrefetcher:
    """`for batch in DevicePrefetcher(loader, trainer): trainer.iter(batch)` -- a background thread takes the host batches of
    `loader` one ahead of the training thread: uploads (pinned, on the copy stream), interning of the index tensors, and
    -- what matters for variable meshes, where every batch is a new block-diagonal graph -- the plans (host CSR builds)
    and the edge-weight chain of the batch.  The training thread then only enqueues the step.  The reference has no
    counterpart (its DataLoader hands CPU batches to `move_to_device`, trainer/trainer.py:143); results are identical to
    feeding `trainer.iter` directly."""

    _END = object()

    def __init__(self, loader, trainer, depth=2):
        self.loader, self.trainer, self.depth = loader, trainer, max(int(depth), 1)

    def __iter__(self):
        import queue
        import threading
        q = queue.Queue(maxsize=self.depth)
        stop = threading.Event()

        def put(item):                 # bounded put that gives up when the consumer has gone away
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    continue
            return False

        def work():
            try:
                torch.cuda.set_device(self.trainer.device)
                for data in self.loader:
                    if not put(self.trainer.prefetch(data)):
                        return
                put(self._END)
            except BaseException as e:     # surfaces in the training thread
                put(e)

        th = threading.Thread(target=work, name="bsms-prefetch", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is self._END:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            stop.set()
            th.join(timeout=5.0)


class Trainer:
    """src/trainer/trainer.py:9-229.  `model_cfg` needs consistent_mesh, accumulation_steps; `opt_cfg` needs
    peak_lr, weight_decay, warmup_steps, decay_steps, gnorm_clip (configs/opt/default.yaml)."""

    def __init__(self, model, model_cfg, opt_cfg):
        self.model_cfg, self.opt_cfg = model_cfg, opt_cfg
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.model = model.to(self.device)
        self.dp = DataParallel(self.model)                       # world size 1: no collective is issued
        self.optimizer = FusedAdamW(self.dp.grads, lr=opt_cfg.peak_lr, weight_decay=opt_cfg.weight_decay,
                                    max_grad_norm=opt_cfg.gnorm_clip)
        self.lr_scheduler = WarmupCosineDecay(opt_cfg.peak_lr, opt_cfg.warmup_steps, opt_cfg.decay_steps)
        self.train_step = 0
        _check_host_threads()
        self._synced = False
        self._norm_base = None     # statistics every rank already shares (restore followed by warm-up)

    def move_to_device(self, data):
        """trainer/trainer.py:143 semantics (nested lists -> device), except that INDEX tensors (edge lists, kept ids)
        are interned by content (graph.intern_index): a batch of the same mesh maps to the same device tensors every
        step, so the per-mesh plans (CSR layouts, edge weights) are built once, not once per step."""
        from .graph import LevelData, intern_index, _upload
        if isinstance(data, (list, tuple)):
            return [self.move_to_device(d) for d in data]
        if isinstance(data, LevelData):
            return data.to(self.device, intern=True)
        if data.dtype == torch.int64 and not data.is_cuda:
            return intern_index(data, self.device, shared_batch_axis=bool(self.model_cfg.consistent_mesh))
        return _upload(data, self.device)

    def collate(self, samples):
        """Variable meshes (`consistent_mesh: false`): the device batch of a list of SAMPLES (each a per-level list of host LevelData,
        what a dataset item is before PyG's `Batch` collation, datasets/base.py:325-349), assembled on the GPU from per-mesh state
        that stays in HBM (graph.MeshBank: the meshes' plans and edge weights are built once, a batch costs one bsms_plan_concat per
        level).  `iter` / `get_loss` / `get_pred` accept the result like any device batch.  Use it as the loader's `collate_fn`
        consumer: `trainer.iter(trainer.collate(samples))` -- a fresh combination of meshes then costs what a cached one does
        (profiles/fresh_mesh.py: 2.4 against 3.9 ms per cylinder batch of 8)."""
        from .graph import MeshBank
        if self.model_cfg.consistent_mesh:
            raise ValueError("Trainer.collate is the variable-mesh path (model_cfg.consistent_mesh = False)")
        if getattr(self, "_bank", None) is None:
            self._bank = MeshBank(self.model.process, self.device)
        return self._bank.collate(samples)

    def prefetch(self, data):
        """move_to_device + everything mesh-dependent a step of this batch will look up (plans, edge weights): what
        DevicePrefetcher runs one batch ahead.  Returns the device batch; `iter` accepts it as it accepts a host batch."""
        data = self.move_to_device(data)
        if self.model_cfg.consistent_mesh:
            node_in, m_gs, m_ids = data[0], [g[0] for g in data[3]], [i[0] for i in data[4]]
            n0 = node_in.shape[1]
        else:
            m_gs = [d.edge_index for d in data]
            m_ids = [data[i].face for i in range(len(m_gs) - 1)]
            n0 = data[0].x.shape[0]
        self.model.process.prepare(m_ids, m_gs, n0, self.device)
        return data

    def _warming_up(self):
        return self.train_step < self.model_cfg.accumulation_steps

    def _model_forward(self, data):
        return self.model(data, self.model_cfg.consistent_mesh, self._warming_up())

    def get_label_mask(self, data):
        if self.model_cfg.consistent_mesh:
            return data[1], data[2]
        return data[0].y, data[0].mask

    def get_pred(self, data):
        return self._model_forward(self.move_to_device(data))

    def get_loss(self, data):
        data = self.move_to_device(data)
        pred = self._model_forward(data)
        tar, mask = self.get_label_mask(data)
        se = (pred - tar) ** 2
        return torch.sqrt((se * mask).sum() / mask.sum() / se.shape[-1])

    def iter(self, data):
        """One training iteration (trainer.py:134-156): statistics only during warm-up, otherwise
        fwd + loss + bwd (+ gradient all-reduce) + clip + AdamW + LR schedule."""
        data = self.move_to_device(data)
        if self._warming_up():
            self._model_forward(data)
            loss = None
        else:
            if not self._synced:                                  # merge normaliser statistics once (dp.py)
                self.dp.sync_normalizers(self._norm_base)
                self._synced, self._norm_base = True, None
            loss = self.dp.step_loss_backward(data, self.model_cfg.consistent_mesh)
            self.optimizer.step(self.lr_scheduler.lr())
            self.lr_scheduler.step()
        self.train_step += 1
        return loss

    def save(self, save_dir):
        os.makedirs(save_dir, exist_ok=True)
        torch.save(self.model.state_dict(), f"{save_dir}/{self.train_step}_params.pth")     # reference layout
        torch.save({"opt": self.optimizer.state_dict(), "epoch": self.lr_scheduler.last_epoch, "train_step": self.train_step},
                   f"{save_dir}/{self.train_step}_opt_state.pth")                           # the reference's TODO

    def restore(self, save_dir, step, restore_opt_state=True):
        self.model.load_state_dict(torch.load(f"{save_dir}/{step}_params.pth", map_location=self.device))
        opt_path = f"{save_dir}/{step}_opt_state.pth"
        if restore_opt_state and os.path.exists(opt_path):
            st = torch.load(opt_path, map_location=self.device)
            self.optimizer.load_state_dict(st["opt"])
            self.lr_scheduler.last_epoch = st["epoch"]
            self.train_step = st["train_step"]
        # Restored normaliser statistics are already the merged ones.  If no warm-up follows they must not be merged
        # again (it would multiply _acc_weight / _num_accumulations by the world size).  If warm-up does follow
        # (restore_opt_state=False, no optimizer-state file, or a checkpoint taken during warm-up) every rank goes on
        # accumulating its own shard on top of them: the merge then runs once after warm-up, over the per-rank
        # additions only, with the restored part counted once (Normalizer.synchronize(base=...)).
        if self._warming_up():
            self._synced, self._norm_base = False, {m: m.snapshot() for m in self.dp.normalizers()}
        else:
            self._synced, self._norm_base = True, None
