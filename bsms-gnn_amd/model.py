"""Drop-in for the reference's model layer (src/models/model.py, src/utils/normalizer.py) and the
loss of src/trainer/trainer.py:96-97.  Encoder, processor (BSGMP) and decoder run on libbsms_hip.so;
the fp64 normaliser arithmetic on [B,N,<=4] tensors and the masked integration are a handful of
elementwise PyTorch-ROCm ops (host plumbing)."""
import torch
from torch import nn

from . import _abi
from .ops import BSGMP, MLP, _stream


class Normalizer(nn.Module):
    """utils/normalizer.py:9-114.  Same state_dict keys/dtypes (fp64) as the reference.

    `synchronize()` is the data-parallel reduction the reference left as dead code
    (normalizer.py:92-114, its call at :37 is commented out): weighted merge of the per-rank
    accumulators so every rank normalises identically."""

    def __init__(self, size, max_accumulations=10**6, std_epsilon=1e-8, unit=10**6, dtype=torch.float64,
                 device="cpu", name="Normalizer"):
        super().__init__()
        self.name, self.unit, self.size, self.dtype, self.synced = name, unit, size, dtype, False
        mk = lambda t: nn.Parameter(t, requires_grad=False)
        self.std_eps = mk(torch.tensor(std_epsilon, dtype=dtype, device=device))
        self._max_accumulations = mk(torch.tensor(max_accumulations, dtype=dtype, device=device))
        self._acc_weight = mk(torch.zeros(1, dtype=dtype, device=device))
        self._num_accumulations = mk(torch.zeros(1, dtype=dtype, device=device))
        self._E_data = mk(torch.zeros(size, dtype=dtype, device=device))
        self._E_data_squared = mk(torch.zeros(size, dtype=dtype, device=device))

    def forward(self, batched_data, accumulate=False):
        if accumulate and bool(self._num_accumulations < self._max_accumulations):
            self._accumulate(batched_data)
        return ((batched_data - self.mean()) / self.std_with_epsilon()).type(torch.float32)

    def _accumulate(self, batched_data):
        rows = batched_data.reshape(-1, self.size)
        old = self._acc_weight.data
        # normalizer.py:58: torch.tensor(python float) is FLOAT32, cast to fp64 afterwards -- the weight of a batch is the fp32
        # rounding of rows / unit (it cancels when all batches have the same size, not otherwise)
        dw = torch.tensor(rows.shape[0] / self.unit).type(self.dtype).to(rows.device)
        m1 = torch.mean(rows, dim=0).type(self.dtype)
        m2 = torch.mean(rows ** 2, dim=0).type(self.dtype)
        self._acc_weight.data = old.add(dw)
        self._E_data.data = self._E_data.data.multiply(old).add(m1.multiply(dw)).divide(self._acc_weight)
        self._E_data_squared.data = self._E_data_squared.data.multiply(old).add(m2.multiply(dw)).divide(self._acc_weight)
        self._num_accumulations.data = self._num_accumulations.data.add(1.0)

    def inverse(self, normalized_batch_data):
        return ((normalized_batch_data * self.std_with_epsilon()) + self.mean()).type(torch.float32)

    def mean(self):
        return self._E_data

    def std_with_epsilon(self):
        std = torch.sqrt(self._E_data_squared - self.mean() ** 2)
        return torch.max(torch.nan_to_num(std), self.std_eps)

    def snapshot(self):
        """(weight, count, weighted first / second moments) of the statistics held now -- the `base` of a later
        `synchronize` when these statistics are already global (restored from a checkpoint)."""
        w = self._acc_weight.data.clone()
        return w, self._num_accumulations.data.clone(), self._E_data.data * w, self._E_data_squared.data * w

    def synchronize(self, group=None, base=None):
        """Merge accumulators across ranks (weights add; means are weight-averaged).  `base` (a `snapshot()`) is a
        part of the statistics every rank already shares -- a restored checkpoint followed by more warm-up -- and
        is counted once instead of once per rank."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            self.synced = True
            return
        w = self._acc_weight.data.clone()
        parts = [w, self._num_accumulations.data.clone(), self._E_data.data * w, self._E_data_squared.data * w]
        if base is not None:
            parts = [a - b for a, b in zip(parts, base)]
        buf = torch.cat(parts)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        if base is not None:
            buf = buf + torch.cat(list(base))
        wsum = buf[0:1]
        self._acc_weight.data = wsum.clone()
        self._num_accumulations.data = buf[1:2].clone()
        safe = torch.where(wsum > 0, wsum, torch.ones_like(wsum))
        self._E_data.data = buf[2:2 + self.size] / safe
        self._E_data_squared.data = buf[2 + self.size:2 + 2 * self.size] / safe
        self.synced = True


class BSMS_Simulator(nn.Module):
    """models/model.py:8-208.  forward(data, consistent_mesh, warmup); cfg needs
    out_dim, latent_dim, hidden_layer, unet_depth, pos_dim (any attribute object)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.encode = MLP(cfg.out_dim + 1, cfg.latent_dim, cfg.latent_dim, cfg.hidden_layer, True)
        self.process = BSGMP(cfg.unet_depth, cfg.latent_dim, cfg.hidden_layer, cfg.pos_dim)
        self.decode = MLP(cfg.latent_dim, cfg.latent_dim, cfg.out_dim, cfg.hidden_layer, False)
        self.pos_dim = cfg.pos_dim
        self._inputNormalizer = Normalizer(cfg.out_dim + 1, max_accumulations=5e5, name="in_norm")
        self._targetNormalizer = Normalizer(cfg.out_dim, max_accumulations=5e5, name="out_norm")

    def _get_nodal_latent_input(self, node_in):  # model.py:29-46
        return torch.cat([node_in[..., : -1 - self.pos_dim], node_in[..., -1:]], dim=-1)

    def _get_pos_type(self, node_in):            # model.py:48-62
        return node_in[..., -(1 + self.pos_dim): -1].contiguous(), node_in[..., -1].clone()

    def _deltas(self, node_in, node_tar):        # model.py:64-81
        return node_tar - node_in[..., : node_tar.shape[-1]]

    def _encode_process_decode(self, node_feature, m_ids, multi_gs, pos):  # model.py:83-106
        x = self.encode(node_feature)
        x = self.process(x, m_ids, multi_gs, pos)
        return self.decode(x)

    def _warmup(self, node_in, node_tar):        # model.py:108-125
        node_in = self._get_nodal_latent_input(node_in)
        self._inputNormalizer(node_in, accumulate=True)
        self._targetNormalizer(self._deltas(node_in, node_tar), accumulate=True)
        return node_tar.new_zeros(node_tar.shape)

    def _forward(self, m_ids, m_gs, node_in, node_mask):  # model.py:127-164
        node_pos, _ = self._get_pos_type(node_in)
        node_in = self._get_nodal_latent_input(node_in)
        norm_in = self._inputNormalizer(node_in, accumulate=False)
        norm_pred = self._encode_process_decode(norm_in, m_ids, m_gs, node_pos)
        pred_delta = self._targetNormalizer.inverse(norm_pred) * node_mask
        return node_in[..., : pred_delta.shape[-1]] + pred_delta

    def _infer(self, m_ids, m_gs, node_in, node_mask, next_in=None, ic=None, session=None):
        """Forward only (no autograd): `_forward` with the normaliser / integration glue as two fused kernels
        (bsms_sim_prologue / bsms_sim_epilogue) instead of ~15 element-wise launches.  `next_in` (may alias `node_in`)
        receives the next autoregressive input where(mask == 0, ic, cat[pred, mesh_pos | type])
        (utils/rollout_utils.py:57-62); `session`: ops.InferenceSession of an autoregressive caller."""
        if not node_in.is_cuda:
            raise _abi.BsmsError("BSMS_Simulator: the BSMS engine runs on the GPU only; there is no CPU fallback")
        L, s = _abi.lib(), _stream()
        node_in = node_in if (node_in.is_contiguous() and node_in.dtype == torch.float32) else node_in.contiguous().float()
        mask = node_mask if (node_mask.is_contiguous() and node_mask.dtype == torch.float32) else node_mask.contiguous().float()
        B, N, W = node_in.shape
        C, p = W - 1 - self.pos_dim, self.pos_dim
        R = B * N
        if mask.numel() != R:
            raise RuntimeError(f"node_mask has {mask.numel()} entries for {R} nodes")
        ni, no = self._inputNormalizer, self._targetNormalizer
        norm_in = torch.empty(B, N, C + 1, device=node_in.device, dtype=torch.float32)
        pos = torch.empty(B, N, p, device=node_in.device, dtype=torch.float32)
        _abi.check(L.bsms_sim_prologue(node_in.data_ptr(), R, C, p, ni._E_data.data_ptr(), ni._E_data_squared.data_ptr(),
                                       ni.std_eps.data_ptr(), norm_in.data_ptr(), pos.data_ptr(), s), "bsms_sim_prologue")
        x = self.encode(norm_in, session=None if session is None else session.encode)
        x = self.process(x, m_ids, m_gs, pos, session=session)
        y = self.decode(x, session=None if session is None else session.decode)
        pred = torch.empty(B, N, C, device=node_in.device, dtype=torch.float32)
        _abi.check(L.bsms_sim_epilogue(y.data_ptr(), node_in.data_ptr(), mask.data_ptr(), None, R, C, p, no._E_data.data_ptr(),
                                       no._E_data_squared.data_ptr(), no.std_eps.data_ptr(), pred.data_ptr(),
                                       None if next_in is None else next_in.data_ptr(), None if ic is None else ic.data_ptr(),
                                       None, None, s), "bsms_sim_epilogue")
        return pred

    def forward(self, data, consistent_mesh, warmup):  # model.py:166-208
        if consistent_mesh:
            node_in, node_tar, node_mask, m_gs, m_ids = data
            m_gs = [g[0] for g in m_gs]
            m_ids = [i[0] for i in m_ids]
        else:  # PyG-Batch-like objects: block-diagonal graph, batch axis of 1
            node_in, node_tar, node_mask = data[0].x.unsqueeze(0), data[0].y.unsqueeze(0), data[0].mask.unsqueeze(0)
            m_gs = [d.edge_index for d in data]
            m_ids = [data[i].face for i in range(len(m_gs) - 1)]
        if warmup:
            return self._warmup(node_in, node_tar)
        if not torch.is_grad_enabled() and node_in.is_cuda and not self.process.per_block:
            return self._infer(m_ids, m_gs, node_in, node_mask)
        return self._forward(m_ids, m_gs, node_in, node_mask)


def masked_rmse(pred, tar, mask):
    """trainer/trainer.py:96-97."""
    se = (pred - tar) ** 2
    return torch.sqrt((se * mask).sum() / mask.sum() / se.shape[-1])
