"""Forward-only autoregressive rollout (reference: src/utils/rollout_utils.py:14-64, src/rollout.py:64-112).

`rollout_one_traj` keeps the reference's signature.  A step is five calls -- prologue kernel, encoder, U-Net (C ABI
`saved = NULL`: no activation is written; weight packs and coarse positions are reused from the previous step, they
cannot change during a rollout), decoder, epilogue kernel -- and the epilogue writes the next input (the reference's
`torch.cat` + `torch.where`) directly into the static input buffer.  Because a consistent mesh makes every step the
same launch sequence on the same buffers, the step can also be captured once into a HIP graph and replayed.
`rollout_batch` advances B trajectories of the same mesh at once (the reference rolls out one at a time); under data
parallelism every rank takes its own slice of the trajectories -- there is nothing to exchange."""
import torch


class _Stepper:
    """One rollout step `cur -> next` on static buffers, optionally replayed from a HIP graph.  The whole step is
    prologue kernel -> encode -> U-Net -> decode -> epilogue kernel: the epilogue writes the prediction AND the next
    input (`torch.cat` + `torch.where` of the reference, rollout_utils.py:57-62) straight into the static input buffer;
    the U-Net reuses its weight packs and the coarse positions from the previous step (ops.InferenceSession)."""

    def __init__(self, model, ic, node_mask, m_gs, m_ids, out_dim, use_graph):
        from .ops import InferenceSession
        self.model, self.c = model, out_dim
        self.ic = ic.contiguous().float()
        self.mask = node_mask.contiguous().float()
        self.m_gs, self.m_ids = [g[0] for g in m_gs], [i[0] for i in m_ids]
        self.cur = self.ic.clone()                    # static input buffer, rewritten in place by every step
        self.session = InferenceSession(static_pos=True)
        self.fused = ic.is_cuda and not model.process.per_block
        self.pred = None
        self.graph = None
        if use_graph and self.fused:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):             # warm-up outside capture: plans, workspaces, packs, coarse positions
                for _ in range(2):
                    self._body()
            torch.cuda.current_stream().wait_stream(side)
            self.cur.copy_(self.ic)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):   # other threads may allocate meanwhile
                self._body()
            self.cur.copy_(self.ic)

    def _body(self):
        if self.fused:
            self.pred = self.model._infer(self.m_ids, self.m_gs, self.cur, self.mask, next_in=self.cur, ic=self.ic,
                                          session=self.session)
            return
        zero = torch.zeros_like(self.ic[..., : self.c])
        pred = self.model((self.cur, zero, self.mask, [g.unsqueeze(0) for g in self.m_gs], [i.unsqueeze(0) for i in self.m_ids]),
                          True, False)
        nxt = torch.cat([pred, self.cur[..., self.c:]], dim=-1)
        self.cur.copy_(torch.where(self.mask == 0, self.ic, nxt))   # Dirichlet nodes keep their IC (rollout_utils.py:62)
        self.pred = pred

    def step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self._body()
        return self.pred


@torch.no_grad()
def rollout_one_traj(trainer, IC, results, node_mask, m_gs, m_ids, cfg=None, use_graph=False):
    """IC [1,N,C+p+1]; results [T-1,N,C] (filled and returned); node_mask [1,N,1]; m_gs/m_ids as the
    consistent-mesh collate delivers them ([1,2,E_l] / [1,N_{l+1}]).  `trainer` is anything with `.model`."""
    model = trainer.model if hasattr(trainer, "model") else trainer
    stepper = _Stepper(model, IC, node_mask, m_gs, m_ids, results.shape[-1], use_graph and IC.is_cuda)
    for ti in range(results.shape[0]):
        results[ti] = stepper.step()[0]
    return results


def rank_slice(n, group=None):
    """[lo, hi) of `n` independent trajectories for this rank: contiguous, sizes differ by at most one.  World size 1 (or
    torch.distributed not initialised): everything."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0, n
    w, r = dist.get_world_size(group), dist.get_rank(group)
    base, extra = divmod(n, w)
    lo = r * base + min(r, extra)
    return lo, lo + base + (1 if r < extra else 0)


@torch.no_grad()
def rollout_batch(trainer, IC, results, node_mask, m_gs, m_ids, use_graph=False, shard=False, group=None, gather=False):
    """B trajectories of one mesh advanced together: IC [B,N,C+p+1]; results [T-1,B,N,C] (filled and returned);
    node_mask [B,N,1]; m_gs / m_ids as the consistent-mesh collate delivers them ([B,2,E_l] / [B,N_{l+1}]).

    `shard=True` under torch.distributed (one process per GPU): trajectories are independent, so rank r advances only
    its slice `rank_slice(B)` of them -- no data-path collective; the other columns of `results`
    are left as they were unless `gather=True`, which all-gathers the slices so that every rank returns the full tensor."""
    model = trainer.model if hasattr(trainer, "model") else trainer
    B = IC.shape[0]
    lo, hi = rank_slice(B, group) if shard else (0, B)
    if hi > lo:
        stepper = _Stepper(model, IC[lo:hi], node_mask[lo:hi], [g[lo:hi] for g in m_gs], [i[lo:hi] for i in m_ids],
                           results.shape[-1], use_graph and IC.is_cuda)
        for ti in range(results.shape[0]):
            results[ti, lo:hi] = stepper.step()
    # rank-independent condition: every rank of the group must enter the collective or none (with B < world a rank's
    # slice can be the whole batch while another's is empty -- deciding from (lo, hi) deadlocked the others)
    import torch.distributed as dist
    if shard and gather and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        w = dist.get_world_size(group)
        width = -(-B // w)                                   # equal-size pieces for all_gather; the tail piece is padded
        mine = results.new_zeros(results.shape[0], width, *results.shape[2:])
        mine[:, : hi - lo] = results[:, lo:hi]
        parts = [torch.empty_like(mine) for _ in range(w)]
        dist.all_gather(parts, mine, group=group)
        base, extra = divmod(B, w)
        o = 0
        for r, part in enumerate(parts):
            n = base + (1 if r < extra else 0)
            results[:, o:o + n] = part[:, :n]
            o += n
    return results


# ------------------------------------------------------------------------------------------------ error statistics
@torch.no_grad()
def rollout_errors(results, target, node_mask):
    """Error figures of ONE rolled-out trajectory exactly as the reference's driver computes them (src/rollout.py:99-107):
    results / target [T-1,N,C], node_mask [T-1,N,1] (the rollout datapipe repeats the node mask over time).
      rmse   [1,1]     sqrt(sum(se * mask) / sum(mask) / C) over all time steps, nodes and channels
      rmse_c [T-1,C]   sqrt(sum_nodes(se * mask) / sum_nodes(mask)) per time step and channel
      rmse_t [C,T-1]   its transpose (what the per-time accumulator is fed)"""
    if node_mask.dim() != 3 or node_mask.shape[0] != results.shape[0] or results.dim() != 3:
        raise ValueError(f"rollout_errors: results / target [T-1,N,C] and node_mask [T-1,N,1] expected, got "
                         f"{tuple(results.shape)} and {tuple(node_mask.shape)} (a [N,1] mask must be repeated over time, "
                         "as the reference's rollout datapipe does)")
    se = (results - target) ** 2
    rmse = torch.sqrt((se * node_mask).sum() / node_mask.sum() / se.shape[-1])
    rmse_c = torch.sqrt((se * node_mask).sum(dim=1) / node_mask.sum(dim=1))
    return rmse.unsqueeze(0).unsqueeze(0), rmse_c, rmse_c.detach().clone().T


def rollout_rmse(results, target, node_mask):
    """Deprecated name of rounds 1-3 (it pooled over time, which is NOT what the reference accumulates): forwards to
    `rollout_errors` -- note the [T-1,N,1] mask and the three return values -- and says so."""
    import warnings
    warnings.warn("rollout_rmse is deprecated: use rollout_errors (mask [T-1,N,1]; returns rmse [1,1], rmse_c [T-1,C], rmse_t [C,T-1])",
                  DeprecationWarning, stacklevel=2)
    return rollout_errors(results, target, node_mask)


class RolloutErrors:
    """The reference driver's three error accumulators (src/rollout.py:64-68, 86-97, 110-112): `Normalizer` instances of
    size 1 / C / T-1 used as running mean and standard deviation over trajectories -- of the overall RMSE, of the
    per-(time, channel) RMSE averaged over time (per channel) and averaged over channels (per time step).
    `synchronize()` merges the accumulators of data-parallel ranks that each rolled out their own trajectories (weights
    add, moments are weight-averaged: `model.Normalizer.synchronize`); `summary()` is what rollout.py:118-160 prints."""

    def __init__(self, device=None):
        self.device = device
        self.all = self.channel = self.time = None

    def add(self, results, target, node_mask):
        from .model import Normalizer
        rmse, rmse_c, rmse_t = rollout_errors(results, target, node_mask)
        if self.all is None:
            dev = results.device if self.device is None else self.device
            self.all = Normalizer(1, device=dev, name="rmse_accumulator")
            self.channel = Normalizer(results.shape[-1], device=dev, name="rmse_accumulators_of_channel")
            self.time = Normalizer(results.shape[0], device=dev, name="rmse_accumulators_of_time")
        self.all(rmse, accumulate=True)
        self.channel(rmse_c, accumulate=True)
        self.time(rmse_t, accumulate=True)
        return rmse, rmse_c, rmse_t

    def _ensure(self, T1, C, device):
        """A rank that got no trajectory still has to take part in the merge (with zero weight)."""
        from .model import Normalizer
        if self.all is None:
            self.all, self.channel, self.time = Normalizer(1, device=device), Normalizer(C, device=device), Normalizer(T1, device=device)

    def synchronize(self, group=None, shape=None, device=None):
        """Merge across ranks.  `shape` = (T-1, C): needed only by a rank that added nothing."""
        if self.all is None:
            if shape is None:
                raise ValueError("RolloutErrors.synchronize: this rank added no trajectory -- pass shape=(T-1, C)")
            self._ensure(shape[0], shape[1], device if device is not None else self.device)
        for a in (self.all, self.channel, self.time):
            a.synchronize(group)
        return self

    def summary(self):
        """{name: (mean, std)} for name in all / channel / time (rollout.py:118-143); fp64 tensors."""
        return {k: (a.mean().detach().clone(), a.std_with_epsilon().detach().clone())
                for k, a in (("all", self.all), ("channel", self.channel), ("time", self.time))}


@torch.no_grad()
def rollout_dataset(trainer, loader, cfg=None, use_graph=False, group=None, errors=None):
    """The loop of the reference's rollout driver (src/rollout.py:69-112) over `loader`, an iterable of rollout batches
    (node_info_inp [1,T-1,N,C+p+1], node_info_tar [1,T-1,N,C], node_mask [1,T-1,N,1], m_gs, m_ids) as its datapipe yields
    them with batch_size=1.  Trajectories are independent: under torch.distributed rank r takes trajectories r, r + world,
    ... (every rank iterates the same loader) and the error accumulators are merged once at the end -- the only
    collective.  Returns the `RolloutErrors` (identical on every rank)."""
    import torch.distributed as dist
    model = trainer.model if hasattr(trainer, "model") else trainer
    dist_on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    world, rank = (dist.get_world_size(group), dist.get_rank(group)) if dist_on else (1, 0)
    dev = next(model.parameters()).device
    errors = RolloutErrors(dev) if errors is None else errors
    shape = None
    for k, batch in enumerate(loader):
        inp, tar, mask, m_gs, m_ids = batch
        shape = (tar.shape[1], tar.shape[-1])
        if k % world != rank:
            continue
        mv = trainer.move_to_device if hasattr(trainer, "move_to_device") else (lambda t: t)
        inp, tar, mask, m_gs, m_ids = mv([inp, tar, mask, m_gs, m_ids])
        inp, tar, mask = inp.squeeze(0), tar.squeeze(0), mask.squeeze(0)                    # rollout.py:76-78
        results = tar.new_zeros(tar.shape)
        results = rollout_one_traj(trainer, inp[0:1].clone(), results, mask[0], m_gs, m_ids, cfg, use_graph=use_graph)   # :81-87 (mask[0]: [N,1] -> broadcasts as the reference's does)
        errors.add(results, tar, mask)
    if dist_on:
        errors.synchronize(group, shape, dev)
    return errors
