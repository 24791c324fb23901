"""CPU oracle for the BSMS-GNN hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file restates, on the CPU and in plain fp32 PyTorch, the arithmetic of the reference's
multi-level message-passing path so the HIP kernels can be checked against it.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it; the product
package (`bsms-gnn_amd/`) never does.

Parity status: PINNED.  Every function here is checked against golden vectors produced by importing
the reference itself in the build container (`tests/golden/make_golden.py` -> `tests/golden/*.npz`,
test `tests/test_oracle_golden.py`).  The reference has no tests of its own,
so those generated vectors are the pin.

Each function cites the reference lines (relative to /root/reference/src) it restates.
The op ORDER follows the reference exactly so results agree to fp32 round-off (mostly bit-exact).
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import List, Optional, Sequence

import torch
from torch import nn

# --------------------------------------------------------------------------------------------
# tensor primitives                                                        utils/basic.py:287-343
# --------------------------------------------------------------------------------------------


def degree(index: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    """Histogram of `index`; length is max(index)+1, NOT num_nodes (utils/basic.py:305-309)."""
    n = int(index.max()) + 1
    hist = torch.zeros(n, dtype=dtype)
    hist.scatter_add_(0, index, torch.ones(index.numel(), dtype=dtype))
    return hist


def scatter_sum(src: torch.Tensor, index: torch.Tensor, dim: int, dim_size: int) -> torch.Tensor:
    """out[..., index[e], ...] += src[..., e, ...] along `dim` (utils/basic.py:324-343).

    `index` is 1-D over `dim`; it is expanded (not copied) to src's shape (utils/basic.py:312-321)."""
    d = dim if dim >= 0 else src.dim() + dim
    shape = [1] * src.dim()
    shape[d] = index.numel()
    idx = index.view(shape).expand_as(src)
    out_shape = list(src.shape)
    out_shape[d] = dim_size
    return torch.zeros(out_shape, dtype=src.dtype).scatter_add_(d, idx, src)


# --------------------------------------------------------------------------------------------
# MLP / GMP / transitions                                                       ops/basic.py
# --------------------------------------------------------------------------------------------


class MLP(nn.Module):
    """hidden_layers x (Linear, ReLU) + Linear (+ LayerNorm without affine)  (ops/basic.py:6-23).

    Parameters live under `seq.{0,2,4,..}.{weight,bias}` exactly like the reference so golden
    state_dicts load unchanged."""

    def __init__(self, input_dim, latent_dim, output_dim, hidden_layers, layer_normalized=True):
        super().__init__()
        widths = [input_dim] + [latent_dim] * hidden_layers
        parts: List[nn.Module] = []
        for fan_in, fan_out in zip(widths[:-1], widths[1:]):
            parts += [nn.Linear(fan_in, fan_out), nn.ReLU()]
        parts.append(nn.Linear(widths[-1], output_dim))
        if layer_normalized:
            parts.append(nn.LayerNorm(output_dim, elementwise_affine=False))
        self.seq = nn.Sequential(*parts)

    def forward(self, x):
        return self.seq(x)


def _take_nodes(t: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    if t.dim() == 3:
        return t[:, idx]
    if t.dim() == 2:
        return t[idx]
    raise NotImplementedError("Only implemented for dim 2 and 3")  # ops/basic.py:74,81,134


class GMP(nn.Module):
    """One message-passing block (ops/basic.py:26-98).

    message_e = mlp_edge([pos_i - pos_j, |pos_i - pos_j|, x_i, x_j]),  i = g[0] (source), j = g[1];
    aggr_j = sum over edges into j;  out = mlp_node([x, aggr]) + x."""

    def __init__(self, latent_dim, hidden_layer, pos_dim):
        super().__init__()
        self.mlp_node = MLP(2 * latent_dim, latent_dim, latent_dim, hidden_layer)
        self.mlp_edge = MLP(2 * latent_dim + pos_dim + 1, latent_dim, latent_dim, hidden_layer)
        self.pos_dim = pos_dim

    def forward(self, x, g, pos):
        send, recv = g[0], g[1]
        x_send, x_recv = _take_nodes(x, send), _take_nodes(x, recv)          # :70-72
        rel = _take_nodes(pos, send) - _take_nodes(pos, recv)                # :77-83
        fiber = torch.cat([rel, torch.norm(rel, dim=-1, keepdim=True)], -1)  # :84-85
        if x.dim() == 3 and pos.dim() == 2:                                  # :87-88
            fiber = fiber.unsqueeze(0).repeat(x.shape[0], 1, 1)
        msg = self.mlp_edge(torch.cat([fiber, x_send, x_recv], -1))          # :90-92
        aggr = scatter_sum(msg, recv, dim=-2, dim_size=x.shape[-2])          # :94
        return self.mlp_node(torch.cat([x, aggr], -1)) + x                   # :97-98


def edge_conv(x, g, ew, aggragating=True):
    """WeightedEdgeConv.forward (ops/basic.py:107-140): out[tgt_e] += ew_e * x[from_e]."""
    send, recv = g[0], g[1]
    frm, tgt = (send, recv) if aggragating else (recv, send)
    carried = _take_nodes(x, frm) * ew.unsqueeze(-1)
    return scatter_sum(carried, tgt, dim=-2, dim_size=x.shape[-2])


@torch.no_grad()
def cal_ew(w, g):
    """WeightedEdgeConv.cal_ew (ops/basic.py:142-167). Returns (ec [E], aggr_w [N] incl. 1e-12)."""
    send, recv = g[0], g[1]
    share = w.squeeze(-1) / degree(send, dtype=torch.float)                 # :159-160 (fp64 w promotes: exact leg)
    sent = share[send]                                                       # :162
    aggr_w = scatter_sum(sent, recv, dim=-1, dim_size=share.size(0)) + 1e-12  # :163-164
    return sent / aggr_w[recv], aggr_w                                       # :165


def unpool(h, pre_node_num, idx):
    """Unpool.forward (ops/basic.py:176-201): zero-fill then write rows `idx`."""
    if h.dim() == 2:
        out = h.new_zeros([pre_node_num, h.shape[-1]])
        out[idx] = h
    else:
        out = h.new_zeros([h.shape[0], pre_node_num, h.shape[-1]])
        out[:, idx] = h
    return out


class BSGMP(nn.Module):
    """Bi-stride U-Net over levels (ops/BSMS.py:8-104). Same module/parameter names as the reference."""

    def __init__(self, unet_depth, latent_dim, hidden_layer, pos_dim):
        super().__init__()
        self.bottom_gmp = GMP(latent_dim, hidden_layer, pos_dim)
        self.down_gmps = nn.ModuleList()
        self.up_gmps = nn.ModuleList()
        self.unet_depth = unet_depth
        for _ in range(unet_depth):  # interleaved construction order matters for seeded init parity
            self.down_gmps.append(GMP(latent_dim, hidden_layer, pos_dim))
            self.up_gmps.append(GMP(latent_dim, hidden_layer, pos_dim))

    def forward(self, h, m_ids, m_gs, pos, trace: Optional[dict] = None):
        skip_h, skip_pos, weights = [], [], []
        w = pos.new_ones((pos.shape[-2], 1))                                   # BSMS.py:64
        for lvl in range(self.unet_depth):                                     # BSMS.py:67-89
            h = self.down_gmps[lvl](h, m_gs[lvl], pos)
            skip_h.append(h)
            skip_pos.append(pos)
            ew, w = cal_ew(w, m_gs[lvl])
            weights.append(ew)
            h = _take_nodes(edge_conv(h, m_gs[lvl], ew), m_ids[lvl])
            pos = _take_nodes(edge_conv(pos, m_gs[lvl], ew), m_ids[lvl])
            w = w[m_ids[lvl]]
            if trace is not None:
                trace.setdefault("h_pooled", []).append(h.detach().clone())
                trace.setdefault("pos_pooled", []).append(pos.detach().clone())
                trace.setdefault("ew", []).append(ew.clone())
        h = self.bottom_gmp(h, m_gs[self.unet_depth], pos)                     # BSMS.py:92
        for step in range(self.unet_depth):                                    # BSMS.py:95-102
            lvl = self.unet_depth - 1 - step
            h = unpool(h, skip_h[lvl].shape[-2], m_ids[lvl])
            h = edge_conv(h, m_gs[lvl], weights[lvl], aggragating=False)
            h = self.up_gmps[step](h, m_gs[lvl], skip_pos[lvl])
            h = h + skip_h[lvl]
        return h


# --------------------------------------------------------------------------------------------
# Normalizer                                                              utils/normalizer.py
# --------------------------------------------------------------------------------------------


class Normalizer(nn.Module):
    """Online mean / mean-of-squares in fp64 (utils/normalizer.py:9-90); state keys as the reference."""

    def __init__(self, size, max_accumulations=10**6, std_epsilon=1e-8, unit=10**6, out_dtype=torch.float32):
        super().__init__()
        f64 = dict(dtype=torch.float64)
        self.size, self.unit = size, unit
        self.out_dtype = out_dtype   # fp32 like the reference; fp64 only for the "exact" leg of the three-way gradient tests
        mk = lambda t: nn.Parameter(t, requires_grad=False)
        self.std_eps = mk(torch.tensor(std_epsilon, **f64))
        self._max_accumulations = mk(torch.tensor(max_accumulations, **f64))
        self._acc_weight = mk(torch.zeros(1, **f64))
        self._num_accumulations = mk(torch.zeros(1, **f64))
        self._E_data = mk(torch.zeros(size, **f64))
        self._E_data_squared = mk(torch.zeros(size, **f64))

    def std_with_epsilon(self):                                               # :88-90
        std = torch.sqrt(self._E_data_squared - self._E_data ** 2)
        return torch.max(torch.nan_to_num(std), self.std_eps)

    def accumulate(self, batch):                                              # :54-71
        rows = batch.view(-1, self.size)
        w_old = self._acc_weight.data
        dw = torch.tensor(rows.shape[0] / self.unit).type(torch.float64)
        m1 = torch.mean(rows, dim=0).type(torch.float64)
        m2 = torch.mean(rows ** 2, dim=0).type(torch.float64)
        self._acc_weight.data = w_old.add(dw)
        self._E_data.data = self._E_data.data.multiply(w_old).add(m1.multiply(dw)).divide(self._acc_weight)
        self._E_data_squared.data = (
            self._E_data_squared.data.multiply(w_old).add(m2.multiply(dw)).divide(self._acc_weight)
        )
        self._num_accumulations.data = self._num_accumulations.data.add(1.0)

    def forward(self, batch, accumulate=False):                               # :40-52
        if accumulate and bool(self._num_accumulations < self._max_accumulations):
            self.accumulate(batch)
        return ((batch - self._E_data) / self.std_with_epsilon()).type(self.out_dtype)

    def inverse(self, batch):                                                 # :80-83
        return ((batch * self.std_with_epsilon()) + self._E_data).type(self.out_dtype)


# --------------------------------------------------------------------------------------------
# Simulator + loss                                        models/model.py, trainer/trainer.py
# --------------------------------------------------------------------------------------------


class BSMS_Simulator(nn.Module):
    """normalise -> encode -> BSGMP -> decode -> de-normalise -> mask -> integrate (models/model.py)."""

    def __init__(self, cfg, dtype=torch.float32):
        """`dtype`: arithmetic of the network.  fp32 = the reference; fp64 (weights cast with .double(), fp64 inputs)
        is the near-exact leg the full-size gradient tests measure both fp32 implementations against."""
        super().__init__()
        self.cfg = cfg
        self.pos_dim = cfg.pos_dim
        self.encode = MLP(cfg.out_dim + 1, cfg.latent_dim, cfg.latent_dim, cfg.hidden_layer, True)
        self.process = BSGMP(cfg.unet_depth, cfg.latent_dim, cfg.hidden_layer, cfg.pos_dim)
        self.decode = MLP(cfg.latent_dim, cfg.latent_dim, cfg.out_dim, cfg.hidden_layer, False)
        self._inputNormalizer = Normalizer(cfg.out_dim + 1, max_accumulations=5e5, out_dtype=dtype)
        self._targetNormalizer = Normalizer(cfg.out_dim, max_accumulations=5e5, out_dtype=dtype)

    def split(self, node_in):
        """node_in[..., :C] state | [..., C:C+p] mesh_pos | [..., -1] node_type (model.py:43-46,62)."""
        p = self.pos_dim
        latent_in = torch.cat([node_in[..., : -1 - p], node_in[..., -1:]], dim=-1)
        return latent_in, node_in[..., -(1 + p): -1]

    def forward(self, data, consistent_mesh=True, warmup=False):
        node_in, node_tar, node_mask, m_gs, m_ids = data
        if consistent_mesh:                                                   # model.py:189-192
            m_gs = [g[0] for g in m_gs]
            m_ids = [i[0] for i in m_ids]
        latent_in, pos = self.split(node_in)
        if warmup:                                                            # model.py:108-125,201-206
            self._inputNormalizer(latent_in, accumulate=True)
            self._targetNormalizer(node_tar - latent_in[..., : node_tar.shape[-1]], accumulate=True)
            return node_tar.new_zeros(node_tar.shape)
        z = self.encode(self._inputNormalizer(latent_in))                     # model.py:153,103
        z = self.decode(self.process(z, m_ids, m_gs, pos))                    # model.py:104-105
        delta = self._targetNormalizer.inverse(z) * node_mask                 # model.py:160-162
        return latent_in[..., : delta.shape[-1]] + delta                      # model.py:163


def masked_rmse(pred, tar, mask):
    """trainer/trainer.py:96-97."""
    se = (pred - tar) ** 2
    return torch.sqrt((se * mask).sum() / mask.sum() / se.shape[-1])


@torch.no_grad()
def rollout(model, initial, node_mask, m_gs, m_ids, steps):
    """Autoregressive forward-only loop (utils/rollout_utils.py:14-64); `initial` is [1,N,C+p+1]."""
    c = model.cfg.out_dim
    cur = initial.clone()
    tail = cur[..., c:].clone()
    frames = []
    for _ in range(steps):
        pred = model((cur, cur.new_zeros(cur.shape), node_mask, m_gs, m_ids), True, False)
        frames.append(pred[0])
        cur = torch.where(node_mask == 0, initial, torch.cat([pred, tail], dim=-1))
    return torch.stack(frames)


def rollout_errors(results, target, node_mask):
    """Per-trajectory error figures of the rollout driver (src/rollout.py:99-107): results / target [T-1,N,C],
    node_mask [T-1,N,1].  Returns rmse [1,1], rmse_c [T-1,C] (RMSE over the nodes per time step and channel) and its
    transpose rmse_t [C,T-1]."""
    se = (results - target) ** 2                                                      # :99
    rmse = torch.sqrt((se * node_mask).sum() / node_mask.sum() / se.shape[-1])        # :101
    rmse_c = torch.sqrt((se * node_mask).sum(dim=1) / node_mask.sum(dim=1))           # :106
    return rmse.unsqueeze(0).unsqueeze(0), rmse_c, rmse_c.detach().clone().T          # :103, :107


class RolloutErrorStats:
    """The three accumulators of src/rollout.py:64-68, 86-97, 110-112 (`Normalizer` instances of size 1 / C / T-1 used as
    running mean / std): `add` one trajectory at a time; mean / std as printed at :118-143."""

    def __init__(self):
        self.all = self.channel = self.time = None

    def add(self, results, target, node_mask):
        rmse, rmse_c, rmse_t = rollout_errors(results, target, node_mask)
        if self.all is None:                                                          # :86-97
            self.all, self.channel, self.time = Normalizer(1), Normalizer(results.shape[-1]), Normalizer(results.shape[0])
        self.all(rmse, accumulate=True)                                               # :110-112
        self.channel(rmse_c, accumulate=True)
        self.time(rmse_t, accumulate=True)
        return rmse, rmse_c, rmse_t


def make_cfg(out_dim, latent_dim, hidden_layer, unet_depth, pos_dim):
    return SimpleNamespace(out_dim=out_dim, latent_dim=latent_dim, hidden_layer=hidden_layer,
                           unet_depth=unet_depth, pos_dim=pos_dim)


# --------------------------------------------------------------------------------------------
# Batch collation (variable meshes)                 datasets/base.py:319-351 + PyG Batch semantics
# --------------------------------------------------------------------------------------------


def collate_block_diagonal(samples: Sequence[dict]):
    """Offset-concatenate graphs into one block-diagonal graph (what PyG `Batch` does to the
    per-level `Data` objects of datasets/base.py:325-349; PyG 2.5.3 is not in the reference tree).

    Each sample: dict(x [N,*], m_gs list of [2,E_l], m_ids list of [N_{l+1}]).
    Level-l edge indices and `m_ids[l]` are shifted by the cumulative node count of level l."""
    depth = len(samples[0]["m_ids"])
    x = torch.cat([s["x"] for s in samples], 0)
    m_gs, m_ids = [], []
    for lvl in range(depth + 1):
        off, gs, ids = 0, [], []
        for s in samples:
            n_l = s["x"].shape[0] if lvl == 0 else s["m_ids"][lvl - 1].numel()
            gs.append(s["m_gs"][lvl] + off)
            if lvl < depth:
                ids.append(s["m_ids"][lvl] + off)
            off += n_l
        m_gs.append(torch.cat(gs, 1))
        if lvl < depth:
            m_ids.append(torch.cat(ids, 0))
    return x, m_gs, m_ids
