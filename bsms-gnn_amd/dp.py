"""Data parallelism for the BSMS path: one process per GPU, full model replica, the batch of
independent meshes sharded across ranks.  Collectives go through
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" for the CPU tests):

  * gradients: ONE flat fp32 buffer (7.7 MB for the airfoil model) whose slices are the parameters'
    `.grad`; it is cut into a few large buckets that are all-reduced (SUM) asynchronously as soon
    as backward has produced every gradient in the bucket, so the ring transfer hides behind the rest of
    backward.  xGMI is point-to-point and a ring all-reduce is bound by one link, hence few, large
    messages rather than per-parameter ones.
  * loss: the masked RMSE (trainer/trainer.py:96-97) is non-linear in the batch, so the two global
    sums (sum se*mask, sum mask) are all-reduced BEFORE backward; local gradients are then exact
    partial derivatives of the global loss and are summed, not averaged.
  * the reference replaced nothing here: it wraps nn.DataParallel (trainer/trainer.py:15-18) and then
    disables it (train.py:16); Normalizer.synchronize is dead code (normalizer.py:92-114).
"""
import torch
import torch.distributed as dist


class GradBuckets:
    """Flat fp32 gradient buffer cut into a few large buckets.  Every parameter owns a slot; the engine's
    backward kernels write weight gradients directly into the slots (ops._grad_targets), gradients that
    arrive from plain autograd are copied in by the hook.  A bucket is all-reduced as soon as all of its
    parameters have their gradient."""

    def __init__(self, params, bucket_bytes=2 << 20, group=None, overlap=True):
        self.group = group
        self.overlap = overlap   # False: every bucket is reduced in finish() (after backward)
        self.params = [p for p in params if p.requires_grad]
        order = list(reversed(self.params))  # roughly the order backward produces them
        total = sum(p.numel() for p in order)
        dev, dt = order[0].device, order[0].dtype
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        self.buckets, self._bucket_of, self._slot = [], {}, {}
        off, start, cur = 0, 0, []
        for p in order:
            n = p.numel()
            self._slot[p] = (off, n)
            p._bsms_grad_slot = (self.flat, off, n)
            p._bsms_slot_used = False
            p.grad = None
            cur.append(p)
            off += n
            if (off - start) * self.flat.element_size() >= bucket_bytes:
                self._close(start, off, cur)
                start, cur = off, []
        if cur:
            self._close(start, off, cur)
        self._pending, self._handles = [len(b["params"]) for b in self.buckets], []
        self._seen, self._launched = set(), set()
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._on_grad)

    def _close(self, start, end, plist):
        idx = len(self.buckets)
        self.buckets.append({"view": self.flat[start:end], "params": list(plist)})
        for p in plist:
            self._bucket_of[p] = idx

    def _on_grad(self, p):
        off, n = self._slot[p]
        if p.grad.data_ptr() != self.flat.data_ptr() + off * self.flat.element_size():
            view = self.flat[off:off + n].view_as(p)   # gradient produced outside the engine: move it in
            view.copy_(p.grad)
            p.grad = view
        b = self._bucket_of[p]
        if p in self._seen:
            # second gradient of the same parameter in one backward (a module used twice, accumulation without zero()):
            # the early reduction of its bucket would miss it -- such a bucket is reduced in finish() instead
            if b in self._launched:
                raise RuntimeError("GradBuckets: a parameter received a second gradient after its bucket was all-reduced; "
                                   "call zero() between backward passes or disable bucket overlap")
            self._pending[b] = -1
            return
        self._seen.add(p)
        if self._pending[b] < 0:
            return
        self._pending[b] -= 1
        if self._pending[b] == 0 and self._world() > 1 and self.overlap:
            self._launched.add(b)
            self._handles.append(dist.all_reduce(self.buckets[b]["view"], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _world(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def zero(self):
        self.flat.zero_()
        for p in self.params:
            p.grad = None
            p._bsms_slot_used = False
        self._pending = [len(b["params"]) for b in self.buckets]
        self._seen.clear()
        self._launched.clear()

    def finish(self):
        """Wait for the in-flight bucket reductions (call after backward)."""
        if self._world() > 1:
            for b, left in enumerate(self._pending):  # buckets not reduced during backward (missing / repeated gradients)
                if b not in self._launched:
                    self._launched.add(b)
                    self._handles.append(dist.all_reduce(self.buckets[b]["view"], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for h in self._handles:
            h.wait()
        self._handles = []

    def clip_(self, max_norm):
        """Global-norm clip on the reduced buffer (identical on every rank; trainer/trainer.py:151)."""
        total = torch.linalg.vector_norm(self.flat)
        self.flat.mul_(torch.clamp(max_norm / (total + 1e-6), max=1.0))
        return total


def global_masked_rmse(pred, tar, mask, group=None):
    """sqrt(sum_global(se*mask) / sum_global(mask) / C): value identical on all ranks, gradient flows
    through the local part only (sum over ranks of these local gradients = gradient of the global loss)."""
    se = (pred - tar) ** 2
    s_loc, m_loc = (se * mask).sum(), mask.sum()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        tot = torch.stack([s_loc.detach(), m_loc.detach()])
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
        s = s_loc + (tot[0] - s_loc.detach())
        m = tot[1]
    else:
        s, m = s_loc, m_loc
    return torch.sqrt(s / m / se.shape[-1])


class DataParallel:
    """Wraps a replica: broadcast of parameters at construction, bucketed gradient all-reduce."""

    def __init__(self, model, bucket_bytes=2 << 20, group=None):
        self.model, self.group = model, group
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, src=0, group=group)
        # The U-Net stays ONE bsms_bsgmp_fwd / _bwd call per step under data parallelism too: eight ranks share one
        # host, and the per-block module tree costs every rank ~3x the enqueue time to buy an overlap worth < 3 % (the
        # whole gradient is 7.7 MB: ~0.1-0.2 ms of ring all-reduce over xGMI against a 6-7 ms step).  Buckets complete
        # as autograd hands the gradients over (decoder first, then the whole U-Net, then the encoder) and are
        # all-reduced asynchronously from there.  A model built with BSGMP.per_block = True (BSMS_PY_BSGMP=1) still
        # gets block-by-block overlap.
        self.grads = GradBuckets(list(model.parameters()), bucket_bytes, group)
        # The step itself: direct C-ABI calls on static buffers (step.FusedStep) when the model is the standard
        # BSMS_Simulator; otherwise (or with BSMS_FUSED_STEP=0) autograd over the drop-in modules.
        import os
        from .step import FusedStep
        self.fused = None
        if os.environ.get("BSMS_FUSED_STEP", "1") == "1" and FusedStep.supports(model) and next(model.parameters()).is_cuda:
            self.fused = FusedStep(model, self.grads, group, use_graph=os.environ.get("BSMS_STEP_GRAPH", "0") == "1")

    def __call__(self, *a, **k):
        return self.model(*a, **k)

    def step_loss_backward(self, data, consistent_mesh=True):
        """One fwd + exact global loss + bwd + gradient reduction.  Returns the (global) loss."""
        if self.fused is not None:
            return self.fused(data, consistent_mesh)
        self.grads.zero()
        pred = self.model(data, consistent_mesh, False)
        loss = global_masked_rmse(pred, data[1] if consistent_mesh else data[0].y.unsqueeze(0),
                                  data[2] if consistent_mesh else data[0].mask.unsqueeze(0), self.group)
        loss.backward()
        self.grads.finish()
        return loss

    def normalizers(self):
        return [m for m in self.model.modules() if hasattr(m, "synchronize") and hasattr(m, "_E_data")]

    def sync_normalizers(self, base=None):
        """`base`: {normaliser: snapshot} of statistics all ranks already share (Trainer.restore before more warm-up)."""
        for m in self.normalizers():
            m.synchronize(self.group, None if base is None else base.get(m))
