"""Forward-only autoregressive rollout (reference: src/utils/rollout_utils.py:14-64, src/rollout.py:64-112).

`rollout_one_traj` keeps the reference's signature.  The per-step forward runs in inference mode (the C
ABI's `saved = NULL`: no activation is written) and, because a consistent mesh makes every step the same
launch sequence on the same buffers, the step can be captured once into a HIP graph and replayed -- the
B = 1 rollout launches ~150 few-microsecond kernels per step.  Measured on MI355X (airfoil, B = 1): eager
563 steps/s, graph replay 493 steps/s -- the launches already keep the GPU busy, so eager is the default."""
import torch


class _Stepper:
    """One rollout step `cur -> next` on static buffers, optionally replayed from a HIP graph."""

    def __init__(self, model, ic, node_mask, m_gs, m_ids, out_dim, use_graph):
        self.model, self.ic, self.mask, self.m_gs, self.m_ids, self.c = model, ic, node_mask, m_gs, m_ids, out_dim
        self.cur = ic.clone()                         # static input buffer
        self.tail = ic[..., out_dim:].clone()         # mesh_pos + node_type never change (rollout_utils.py:46)
        self.zero_tar = torch.zeros_like(ic)
        self.pred = None
        self.graph = None
        if use_graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):             # warm-up outside capture: plans, workspaces, allocator
                for _ in range(2):
                    self._body()
            torch.cuda.current_stream().wait_stream(side)
            self.cur.copy_(ic)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._body()
            self.cur.copy_(ic)

    def _body(self):
        pred = self.model((self.cur, self.zero_tar, self.mask, self.m_gs, self.m_ids), True, False)
        nxt = torch.cat([pred, self.tail], dim=-1)
        nxt = torch.where(self.mask == 0, self.ic, nxt)          # Dirichlet nodes keep their IC (rollout_utils.py:62)
        self.pred = pred
        self.cur.copy_(nxt)

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
def rollout_rmse(pred, truth, node_mask):
    """Masked RMSE overall / per channel / per time step (src/rollout.py:99-112)."""
    se = (pred - truth) ** 2 * node_mask
    denom = node_mask.sum() * pred.shape[0]
    overall = torch.sqrt(se.sum() / denom / pred.shape[-1])
    per_channel = torch.sqrt(se.sum(dim=(0, 1)) / denom)
    per_time = torch.sqrt(se.sum(dim=(1, 2)) / node_mask.sum() / pred.shape[-1])
    return overall, per_channel, per_time
