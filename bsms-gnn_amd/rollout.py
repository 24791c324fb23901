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
            with torch.cuda.graph(self.graph):
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


@torch.no_grad()
def rollout_batch(trainer, IC, results, node_mask, m_gs, m_ids, use_graph=False):
    """B trajectories of one mesh advanced together: IC [B,N,C+p+1]; results [T-1,B,N,C] (filled and returned);
    node_mask [B,N,1]; m_gs / m_ids as the consistent-mesh collate delivers them ([B,2,E_l] / [B,N_{l+1}])."""
    model = trainer.model if hasattr(trainer, "model") else trainer
    stepper = _Stepper(model, IC, node_mask, m_gs, m_ids, results.shape[-1], use_graph and IC.is_cuda)
    for ti in range(results.shape[0]):
        results[ti] = stepper.step()
    return results


@torch.no_grad()
def rollout_rmse(pred, truth, node_mask):
    """Masked RMSE overall / per channel / per time step (src/rollout.py:99-112)."""
    se = (pred - truth) ** 2 * node_mask
    denom = node_mask.sum() * pred.shape[0]
    overall = torch.sqrt(se.sum() / denom / pred.shape[-1])
    per_channel = torch.sqrt(se.sum(dim=(0, 1)) / denom)
    per_time = torch.sqrt(se.sum(dim=(1, 2)) / node_mask.sum() / pred.shape[-1])
    return overall, per_channel, per_time
