"""Drop-in modules for the reference's `src/ops` layer, running on libbsms_hip.so (HIP, gfx950).

Same constructor / forward signatures and the same `state_dict` keys as the reference
(/root/reference/src/ops/basic.py, src/ops/BSMS.py, src/utils/basic.py:287-343), so checkpoints and
call sites carry over unchanged:

    MLP(input_dim, latent_dim, output_dim, hidden_layers, layer_normalized=True).forward(x)
    GMP(latent_dim, hidden_layer, pos_dim).forward(x, g, pos)
    WeightedEdgeConv().forward(x, g, ew, aggragating=True) / .cal_ew(w, g)
    Unpool().forward(h, pre_node_num, idx)
    BSGMP(unet_depth, latent_dim, hidden_layer, pos_dim).forward(h, m_ids, m_gs, pos)
    scatter_sum(src, index, dim, out, dim_size), degree(index, num_nodes, dtype)

PyTorch is used for device memory, streams and autograd glue only; every tensor op of the path is a
call through the C ABI.  There is no CPU fallback: CPU tensors raise."""
import os
from typing import Optional

import torch
from torch import nn

from . import _abi
from .graph import LevelPlan, plan_for, plans_for

__all__ = ["MLP", "GMP", "WeightedEdgeConv", "Unpool", "BSGMP", "InferenceSession", "scatter_sum", "degree"]


# ------------------------------------------------------------------------------------ plumbing
def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev_f32(t, what):
    if not t.is_cuda:
        raise _abi.BsmsError(f"{what}: the BSMS engine runs on the GPU only (got a {t.device} tensor); "
                             "there is no CPU fallback")
    if t.dtype != torch.float32:
        raise _abi.BsmsError(f"{what}: float32 expected, got {t.dtype}")
    return t.contiguous()


_WORK = {}
# BSMS_PY_BSGMP=1: run the U-Net as the Python module tree of the reference (one autograd node per block / transition)
# instead of the single bsms_bsgmp_fwd / _bwd call; same kernels, same results, ~3x the host time per step.
_PY_BSGMP = os.environ.get("BSMS_PY_BSGMP", "0") == "1"


def _workspace(device, nbytes):
    """Grow-only scratch per (device, stream); kernels of one stream are ordered, so sharing is safe."""
    key = (device, _stream())
    buf = _WORK.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WORK[key] = buf
    return buf


def _param_ptrs(params):
    return _abi.ptr_array([p.data_ptr() for p in params])


def _grad_targets(params):
    """Where backward writes each weight gradient.  If a data-parallel GradBuckets owns the parameter
    (dp.py), the kernel writes straight into the parameter's slot of the flat all-reduce buffer: autograd
    then adopts that view as `.grad` without a copy or an accumulate kernel.  A second use of the same
    parameter within one step falls back to a fresh tensor (autograd adds it)."""
    out = []
    for p in params:
        slot = getattr(p, "_bsms_grad_slot", None)
        if slot is not None and p.grad is None and not p._bsms_slot_used:
            flat, off, n = slot
            p._bsms_slot_used = True
            out.append(flat[off:off + n].view_as(p))  # fresh view object: nothing else references it
        else:
            out.append(torch.empty_like(p))
    return out


# --------------------------------------------------------------------------------- tensor prims
class _SegmentSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, plan: LevelPlan):  # src [B,E,D] in the caller's edge order
        B, E, D = src.shape
        out = torch.empty(B, plan.N, D, device=src.device, dtype=src.dtype)
        _abi.check(_abi.lib().bsms_segment_sum_fwd(plan.handle, src.data_ptr(), B, D, 0, out.data_ptr(), _stream()),
                   "bsms_segment_sum_fwd")
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, grad):
        grad = grad.contiguous()
        B, N, D = grad.shape
        gsrc = torch.empty(B, ctx.plan.E, D, device=grad.device, dtype=grad.dtype)
        _abi.check(_abi.lib().bsms_segment_sum_bwd(ctx.plan.handle, grad.data_ptr(), B, D, gsrc.data_ptr(), _stream()),
                   "bsms_segment_sum_bwd")
        return gsrc, None


def scatter_sum(src: torch.Tensor, index: torch.Tensor, dim: int = -1, out: Optional[torch.Tensor] = None,
                dim_size: Optional[int] = None) -> torch.Tensor:
    """utils/basic.py:324-343.  `index` is the 1-D target list along `dim`; supported layouts are the
    ones the path uses: src [E] (dim=-1), [E,D] / [B,E,D] (dim=-2)."""
    if out is not None:
        raise NotImplementedError("scatter_sum(out=...) is not used on the BSMS path")
    src = _dev_f32(src, "scatter_sum")
    nd = src.dim()
    d = dim if dim >= 0 else nd + dim
    if index.dim() != 1 or not ((nd == 1 and d == 0) or (nd in (2, 3) and d == nd - 2)):
        raise NotImplementedError("scatter_sum: only a 1-D index along the edge axis ([E], [E,D], [B,E,D]) is implemented")
    if index.numel() != src.shape[d]:
        raise RuntimeError(f"scatter_sum: index has {index.numel()} entries, src has {src.shape[d]} along dim {dim}")
    if dim_size is None:
        dim_size = 0 if index.numel() == 0 else int(index.max()) + 1
    plan = _index_plan(index, dim_size)
    if nd == 1:
        return _SegmentSum.apply(src.view(1, -1, 1), plan).view(dim_size)
    if nd == 2:
        return _SegmentSum.apply(src.unsqueeze(0), plan).squeeze(0)
    return _SegmentSum.apply(src, plan)


_INDEX_PLANS = {}


def _index_plan(index, dim_size):
    """Plan for a bare target index (both COO rows = index); cached on the tensor's storage."""
    from .graph import _key
    key = (_key(index), int(dim_size))
    hit = _INDEX_PLANS.get(key)
    if hit is None:
        if len(_INDEX_PLANS) > 256:
            _INDEX_PLANS.clear()
        idx = index.detach().to(torch.int64)
        hit = (LevelPlan(torch.stack([idx, idx]), dim_size, device=index.device), index.untyped_storage())
        _INDEX_PLANS[key] = hit
    return hit[0]


def degree(index: torch.Tensor, num_nodes: Optional[int] = None, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """utils/basic.py:287-309 -- note the reference IGNORES num_nodes: length is max(index)+1."""
    n = int(index.max()) + 1
    ones = torch.ones(index.numel(), device=index.device, dtype=torch.float32)
    out = scatter_sum(ones, index, dim=-1, dim_size=n)
    return out if dtype in (None, torch.float32) else out.to(dtype)


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t.requires_grad for t in tensors)


# ------------------------------------------------------------------------------------------ MLP
def _mlp_infer(x, hidden, layer_norm, out_dim, params, session=None):
    """Forward only (`saved` = NULL at the ABI): nothing is written for a backward.  `session` (MLPSession): a caller that
    applies the same MLP again and again keeps a private work buffer; the weight packs in it are reused while the
    parameters are unchanged (BSMS_MLP_REUSE_PACKS)."""
    R, in_dim = x.shape
    D = params[0].shape[0]
    L = _abi.lib()
    y = torch.empty(R, out_dim, device=x.device, dtype=x.dtype)
    nbytes = L.bsms_mlp_work_bytes(R, in_dim, D, out_dim, hidden)
    flags = 0
    if session is not None:
        flags = session.flags(params, (R, in_dim, D, out_dim, hidden, int(layer_norm)), nbytes, x.device)
        work = session.work
    else:
        work = _workspace(x.device, nbytes)
    pp, keep = _param_ptrs(params)
    _abi.check(L.bsms_mlp_fwd_ex(x.data_ptr(), R, in_dim, D, out_dim, hidden, int(layer_norm), pp, y.data_ptr(), None,
                                 work.data_ptr(), flags, _stream()), "bsms_mlp_fwd(inference)")
    return y


class MLPSession:
    """Private work buffer + pack validity key of an MLP applied repeatedly in inference (the encoder / decoder of a
    rollout).  Same validity rules as InferenceSession: parameter pointers, autograd versions and the engine's parameter
    epoch (raw-pointer optimizer updates)."""

    def __init__(self):
        self.work, self.key = None, None

    def invalidate(self):
        self.key = None

    def flags(self, params, geom, nbytes, device):
        if self.work is None or self.work.numel() < nbytes or self.work.device != device:
            self.work = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self.key = None
        key = (_PARAM_EPOCH[0], tuple((q.data_ptr(), q._version) for q in params), tuple(geom))
        reuse = 1 if key == self.key else 0       # BSMS_MLP_REUSE_PACKS
        self.key = key
        return reuse


class _MLPFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, hidden, layer_norm, out_dim, *params):
        R, in_dim = x.shape
        D = params[0].shape[0]
        L = _abi.lib()
        y = torch.empty(R, out_dim, device=x.device, dtype=x.dtype)
        saved = torch.empty(L.bsms_mlp_saved_bytes(R, in_dim, D, out_dim, hidden), dtype=torch.uint8, device=x.device)
        work = _workspace(x.device, L.bsms_mlp_work_bytes(R, in_dim, D, out_dim, hidden))
        pp, keep = _param_ptrs(params)
        _abi.check(L.bsms_mlp_fwd(x.data_ptr(), R, in_dim, D, out_dim, hidden, int(layer_norm), pp, y.data_ptr(),
                                  saved.data_ptr(), work.data_ptr(), _stream()), "bsms_mlp_fwd")
        ctx.save_for_backward(x, saved)
        ctx.params = params
        ctx.cfg = (hidden, layer_norm, out_dim)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, saved = ctx.saved_tensors
        params = ctx.params
        hidden, layer_norm, out_dim = ctx.cfg
        R, in_dim = x.shape
        D = params[0].shape[0]
        L = _abi.lib()
        gy = gy.contiguous()
        need_dx = ctx.needs_input_grad[0]
        gx = torch.empty_like(x) if (need_dx or in_dim == D) else None
        grads = _grad_targets(params)
        work = _workspace(x.device, L.bsms_mlp_work_bytes(R, in_dim, D, out_dim, hidden))
        pp, keep = _param_ptrs(params)
        gp, keep2 = _param_ptrs(grads)
        _abi.check(L.bsms_mlp_bwd(x.data_ptr(), gy.data_ptr(), R, in_dim, D, out_dim, hidden, int(layer_norm), pp,
                                  saved.data_ptr(), work.data_ptr(), gx.data_ptr() if gx is not None else None, gp,
                                  _stream()), "bsms_mlp_bwd")
        return (gx if need_dx else None, None, None, None, *grads)


class MLP(nn.Module):
    """ops/basic.py:6-23.  Parameters live in `seq.{0,2,4,...}.{weight,bias}` like the reference."""

    def __init__(self, input_dim, latent_dim, output_dim, hidden_layers, layer_normalized=True):
        super().__init__()
        mods = []
        for l in range(hidden_layers):
            mods += [nn.Linear(input_dim if l == 0 else latent_dim, latent_dim), nn.ReLU()]
        mods.append(nn.Linear(latent_dim, output_dim))
        if layer_normalized:
            mods.append(nn.LayerNorm(output_dim, elementwise_affine=False))
        self.seq = nn.Sequential(*mods)  # parameter container only; the math runs in HIP
        self.hidden_layers, self.layer_normalized = hidden_layers, layer_normalized
        self.input_dim, self.latent_dim, self.output_dim = input_dim, latent_dim, output_dim

    def flat_params(self):
        out = []
        for m in self.seq:
            if isinstance(m, nn.Linear):
                out += [m.weight, m.bias]
        return out

    def forward(self, x, session=None):
        x = _dev_f32(x, "MLP")
        lead = x.shape[:-1]
        params = self.flat_params()
        x2 = x.reshape(-1, x.shape[-1])
        if _needs_grad(x, *params):
            y = _MLPFunction.apply(x2, self.hidden_layers, self.layer_normalized, self.output_dim, *params)
        else:
            y = _mlp_infer(x2, self.hidden_layers, self.layer_normalized, self.output_dim, params, session)
        return y.view(*lead, self.output_dim)


# ------------------------------------------------------------------------------------------ GMP
def _check_gmp_shapes(who, x, pos, plan, latent_dim, pos_dim):
    """The C ABI takes raw pointers and sizes: what the reference would report as a PyTorch shape error must be caught
    here, before it becomes an out-of-bounds device access."""
    if x.shape[-1] != latent_dim:
        raise RuntimeError(f"{who}: feature width {x.shape[-1]} does not match latent_dim {latent_dim}")
    if pos.shape[-1] != pos_dim:
        raise RuntimeError(f"{who}: position width {pos.shape[-1]} does not match pos_dim {pos_dim}")
    if x.shape[-2] != plan.N or pos.shape[-2] != plan.N:
        raise RuntimeError(f"{who}: x has {x.shape[-2]} and pos {pos.shape[-2]} nodes, the graph has {plan.N}")
    if pos.dim() == 3 and x.dim() == 3 and pos.shape[0] != x.shape[0]:
        raise RuntimeError(f"{who}: batch of pos ({pos.shape[0]}) differs from batch of x ({x.shape[0]})")


def _gmp_infer(x, pos, plan, hidden, params):
    """Forward only (`saved` = NULL at the ABI): used by rollout / evaluation under torch.no_grad()."""
    B, N, D = x.shape
    p = pos.shape[-1]
    L = _abi.lib()
    out = torch.empty_like(x)
    work = _workspace(x.device, L.bsms_gmp_work_bytes(B, N, plan.E, D, hidden))
    pp, keep = _param_ptrs(params)
    _abi.check(L.bsms_gmp_fwd(plan.handle, x.data_ptr(), pos.data_ptr(), B, D, p, N * p if pos.dim() == 3 else 0, hidden, pp,
                              out.data_ptr(), None, work.data_ptr(), _stream()), "bsms_gmp_fwd(inference)")
    return out


class _GMPFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos, plan: LevelPlan, hidden, *params):
        B, N, D = x.shape
        p = pos.shape[-1]
        pos_bstride = N * p if pos.dim() == 3 else 0
        L = _abi.lib()
        out = torch.empty_like(x)
        saved = torch.empty(L.bsms_gmp_saved_bytes(B, N, plan.E, D, hidden), dtype=torch.uint8, device=x.device)
        work = _workspace(x.device, L.bsms_gmp_work_bytes(B, N, plan.E, D, hidden))
        pp, keep = _param_ptrs(params)
        _abi.check(L.bsms_gmp_fwd(plan.handle, x.data_ptr(), pos.data_ptr(), B, D, p, pos_bstride, hidden, pp,
                                  out.data_ptr(), saved.data_ptr(), work.data_ptr(), _stream()), "bsms_gmp_fwd")
        ctx.save_for_backward(x, pos, saved)
        ctx.params = params
        ctx.plan, ctx.hidden = plan, hidden
        return out

    @staticmethod
    def backward(ctx, gout):
        x, pos, saved = ctx.saved_tensors
        params = ctx.params
        plan, hidden = ctx.plan, ctx.hidden
        B, N, D = x.shape
        p = pos.shape[-1]
        pos_bstride = N * p if pos.dim() == 3 else 0
        L = _abi.lib()
        gout = gout.contiguous()
        gx = torch.empty_like(x)
        grads = _grad_targets(params)
        work = _workspace(x.device, L.bsms_gmp_work_bytes(B, N, plan.E, D, hidden))
        pp, keep = _param_ptrs(params)
        gp, keep2 = _param_ptrs(grads)
        _abi.check(L.bsms_gmp_bwd(plan.handle, x.data_ptr(), pos.data_ptr(), gout.data_ptr(), B, D, p, pos_bstride,
                                  hidden, pp, saved.data_ptr(), work.data_ptr(), gx.data_ptr(), gp, _stream()),
                   "bsms_gmp_bwd")
        return (gx, None, None, None, *grads)


class GMP(nn.Module):
    """ops/basic.py:26-98.  forward(x, g, pos): x [B,N,C] or [N,C]; g [2,E] int64; pos [B,N,p] or [N,p]."""

    def __init__(self, latent_dim, hidden_layer, pos_dim):
        super().__init__()
        self.mlp_node = MLP(2 * latent_dim, latent_dim, latent_dim, hidden_layer)
        self.mlp_edge = MLP(2 * latent_dim + pos_dim + 1, latent_dim, latent_dim, hidden_layer)
        self.pos_dim, self.hidden_layer, self.latent_dim = pos_dim, hidden_layer, latent_dim

    def forward(self, x, g, pos, plan: Optional[LevelPlan] = None):
        if x.dim() not in (2, 3) or pos.dim() not in (2, 3):
            raise NotImplementedError("Only implemented for dim 2 and 3")
        x = _dev_f32(x, "GMP")
        pos = _dev_f32(pos, "GMP")
        squeeze = x.dim() == 2
        if squeeze:
            if pos.dim() == 3:
                raise NotImplementedError("GMP: 2-D x with 3-D pos is not a layout of the reference")
            x = x.unsqueeze(0)
        if plan is None:
            plan = plan_for(g, x.shape[-2])
        _check_gmp_shapes("GMP", x, pos, plan, self.latent_dim, self.pos_dim)
        params = [*self.mlp_node.flat_params(), *self.mlp_edge.flat_params()]
        if _needs_grad(x, *params):
            y = _GMPFunction.apply(x, pos, plan, self.hidden_layer, *params)
        else:
            y = _gmp_infer(x, pos, plan, self.hidden_layer, params)
        return y.squeeze(0) if squeeze else y


# ---------------------------------------------------------------------------------- transitions
class _EdgeConvFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ew, plan: LevelPlan, aggregating, pooled):
        B, _, D = x.shape
        rows = plan.Nk if (aggregating and pooled) else plan.N
        out = torch.empty(B, rows, D, device=x.device, dtype=x.dtype)
        _abi.check(_abi.lib().bsms_edge_conv(plan.handle, x.data_ptr(), B, D, ew.data_ptr(), int(aggregating), int(pooled),
                                             out.data_ptr(), _stream()), "bsms_edge_conv")
        ctx.save_for_backward(ew)
        ctx.cfg = (plan, aggregating, pooled, x.shape[1])
        return out

    @staticmethod
    def backward(ctx, gout):
        (ew,) = ctx.saved_tensors
        plan, aggregating, pooled, n_in = ctx.cfg
        gout = gout.contiguous()
        B, _, D = gout.shape
        gx = torch.empty(B, n_in, D, device=gout.device, dtype=gout.dtype)
        _abi.check(_abi.lib().bsms_edge_conv(plan.handle, gout.data_ptr(), B, D, ew.data_ptr(), int(not aggregating),
                                             int(pooled), gx.data_ptr(), _stream()), "bsms_edge_conv(adjoint)")
        return gx, None, None, None, None


def _edge_conv(x, ew, plan, aggregating, pooled):
    squeeze = x.dim() == 2
    y = _EdgeConvFunction.apply(x.unsqueeze(0) if squeeze else x, ew, plan, aggregating, pooled)
    return y.squeeze(0) if squeeze else y


class WeightedEdgeConv(nn.Module):
    """ops/basic.py:101-167."""

    def __init__(self, *args):
        super().__init__()

    def forward(self, x, g, ew, aggragating=True, plan: Optional[LevelPlan] = None):
        if x.dim() not in (2, 3):
            raise NotImplementedError("Only implemented for dim 2 and 3")
        x = _dev_f32(x, "WeightedEdgeConv")
        ew = _dev_f32(ew, "WeightedEdgeConv")
        if plan is None:
            plan = plan_for(g, x.shape[-2])
        if ew.numel() != plan.E:
            raise RuntimeError(f"WeightedEdgeConv: {ew.numel()} edge weights for {plan.E} edges")
        if x.shape[-2] != plan.N:
            raise RuntimeError(f"WeightedEdgeConv: x has {x.shape[-2]} nodes, the graph has {plan.N}")
        return _edge_conv(x, ew, plan, bool(aggragating), False)

    @torch.no_grad()
    def cal_ew(self, w, g, plan: Optional[LevelPlan] = None):
        w = _dev_f32(w, "cal_ew")
        n = w.shape[0]
        if plan is None:
            plan = plan_for(g, n)
        if plan.max_source + 1 != n:  # degree() has length max(g[0])+1 (utils/basic.py:305): w / deg would not broadcast
            raise RuntimeError(f"The size of tensor a ({n}) must match the size of tensor b ({plan.max_source + 1}) "
                               "at non-singleton dimension 0")
        w1 = w.reshape(-1).contiguous()
        ec = torch.empty(plan.E, device=w.device, dtype=torch.float32)
        aggr_w = torch.empty(n, device=w.device, dtype=torch.float32)
        _abi.check(_abi.lib().bsms_cal_ew(plan.handle, w1.data_ptr(), ec.data_ptr(), aggr_w.data_ptr(), _stream()),
                   "bsms_cal_ew")
        return ec, aggr_w


class _ScatterRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, idx, n):
        B, nk, D = h.shape
        out = torch.empty(B, n, D, device=h.device, dtype=h.dtype)
        _abi.check(_abi.lib().bsms_scatter_rows(h.data_ptr(), B, nk, D, idx.data_ptr(), n, out.data_ptr(), _stream()),
                   "bsms_scatter_rows")
        ctx.save_for_backward(idx)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, gout):
        (idx,) = ctx.saved_tensors
        gout = gout.contiguous()
        B, n, D = gout.shape
        gh = torch.empty(B, idx.numel(), D, device=gout.device, dtype=gout.dtype)
        _abi.check(_abi.lib().bsms_gather_rows(gout.data_ptr(), B, n, D, idx.data_ptr(), idx.numel(), gh.data_ptr(),
                                               _stream()), "bsms_gather_rows")
        return gh, None, None


_RANGE_OK = {}


def _check_index_range(idx, n):
    """0 <= idx < n, validated ONCE per index tensor (storage identity + version): the check needs a device sync."""
    from .graph import _key
    key = (_key(idx), n)
    if key in _RANGE_OK:
        return
    if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= n):
        raise IndexError(f"Unpool: index out of range for {n} rows")
    if len(_RANGE_OK) > 1024:
        _RANGE_OK.clear()
    _RANGE_OK[key] = idx.untyped_storage()   # keeps the address from being recycled for another tensor


class Unpool(nn.Module):
    """ops/basic.py:170-201."""

    def __init__(self, *args):
        super().__init__()

    def forward(self, h, pre_node_num, idx):
        h = _dev_f32(h, "Unpool")
        idx = idx.to(torch.int64).contiguous()
        if idx.numel() != h.shape[-2]:
            raise RuntimeError(f"Unpool: {idx.numel()} indices for {h.shape[-2]} rows")
        _check_index_range(idx, int(pre_node_num))
        if h.dim() == 2:
            return _ScatterRows.apply(h.unsqueeze(0), idx, int(pre_node_num)).squeeze(0)
        if h.dim() == 3:
            return _ScatterRows.apply(h, idx, int(pre_node_num))
        return None  # the reference falls through for other ranks (ops/basic.py:194-201)


# ---------------------------------------------------------------------------------------- BSGMP
class _BSGMPFunction(torch.autograd.Function):
    """The whole U-Net as ONE autograd node (bsms_bsgmp_fwd / _bwd): two library calls per training step instead of
    ~60 Python autograd nodes."""

    @staticmethod
    def forward(ctx, h, pos, plans, ews, hidden, prec, *params):
        B, _, D = h.shape
        p = pos.shape[-1]
        pos_bstride = pos.shape[-2] * p if pos.dim() == 3 else 0
        L = _abi.lib()
        depth = len(plans) - 1
        pl, keep_pl = _abi.ptr_array([q.handle.value if hasattr(q.handle, "value") else q.handle for q in plans])
        ewp, keep_ew = _abi.ptr_array([e.data_ptr() for e in ews])
        out = torch.empty_like(h)
        saved = torch.empty(L.bsms_bsgmp_saved_bytes_p(pl, depth, B, D, p, hidden, prec), dtype=torch.uint8, device=h.device)
        work = _workspace(h.device, L.bsms_bsgmp_work_bytes(pl, depth, B, D, p, hidden))
        pp, keep = _param_ptrs(params)
        _abi.check(L.bsms_bsgmp_fwd_p(pl, ewp, depth, h.data_ptr(), pos.data_ptr(), B, D, p, pos_bstride, hidden, pp,
                                      out.data_ptr(), saved.data_ptr(), work.data_ptr(), 0, prec, _stream()), "bsms_bsgmp_fwd")
        ctx.save_for_backward(h, pos, saved, *ews)
        ctx.params, ctx.plans, ctx.hidden, ctx.prec = params, plans, hidden, prec
        return out

    @staticmethod
    def backward(ctx, gout):
        h, pos, saved, *ews = ctx.saved_tensors
        params, plans, hidden = ctx.params, ctx.plans, ctx.hidden
        B, _, D = h.shape
        p = pos.shape[-1]
        pos_bstride = pos.shape[-2] * p if pos.dim() == 3 else 0
        L = _abi.lib()
        depth = len(plans) - 1
        pl, keep_pl = _abi.ptr_array([q.handle.value if hasattr(q.handle, "value") else q.handle for q in plans])
        ewp, keep_ew = _abi.ptr_array([e.data_ptr() for e in ews])
        gout = gout.contiguous()
        gh = torch.empty_like(h)
        grads = _grad_targets(params)
        work = _workspace(h.device, L.bsms_bsgmp_work_bytes(pl, depth, B, D, p, hidden))
        pp, keep = _param_ptrs(params)
        gp, keep2 = _param_ptrs(grads)
        _abi.check(L.bsms_bsgmp_bwd_p(pl, ewp, depth, h.data_ptr(), pos.data_ptr(), gout.data_ptr(), B, D, p, pos_bstride, hidden,
                                      pp, saved.data_ptr(), work.data_ptr(), gh.data_ptr(), gp, ctx.prec, _stream()), "bsms_bsgmp_bwd")
        return (gh, None, None, None, None, None, *grads)


_PARAM_EPOCH = [0]


def bump_param_epoch():
    """Called by everything that updates parameters through raw pointers at the C ABI (trainer.FusedAdamW.step): such
    writes do not touch the tensors' autograd version counters, so cached weight packs are keyed on this epoch too."""
    _PARAM_EPOCH[0] += 1


class InferenceSession:
    """State an autoregressive caller keeps between forward-only BSGMP calls (bsms_bsgmp_fwd_ex `reuse`): a PRIVATE work
    buffer holding the weight packs and the coarse positions of the previous call.  Packs are reused while the
    parameters are unchanged (data pointers + version counters + the engine's parameter epoch, which the fused
    optimizer bumps); positions only if the caller declares them static (rollout: mesh_pos never changes,
    utils/rollout_utils.py:46) and the call has the same plans, batch and model geometry (the offsets inside the work
    buffer depend on D / hidden / pos_dim).  `invalidate()` forgets both -- for callers that change parameters or
    positions behind the engine's back."""

    def __init__(self, static_pos=False):
        self.static_pos, self.work, self.pkey, self.gkey = static_pos, None, None, None
        self.encode, self.decode = MLPSession(), MLPSession()    # the encoder / decoder around the U-Net (models/model.py:103-105)

    def invalidate(self):
        self.pkey = self.gkey = None
        self.encode.invalidate()
        self.decode.invalidate()

    def flags(self, params, plans, B, nbytes, device, geom=()):
        if self.work is None or self.work.numel() < nbytes or self.work.device != device:
            self.work = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self.pkey = self.gkey = None
        pkey = (_PARAM_EPOCH[0], tuple((q.data_ptr(), q._version) for q in params))
        gkey = (tuple(q.uid for q in plans), B, tuple(geom))
        reuse = (1 if pkey == self.pkey and gkey == self.gkey else 0) | (2 if self.static_pos and gkey == self.gkey else 0)
        self.pkey, self.gkey = pkey, gkey
        return reuse


PRECISIONS = {"f32": 0, "bf16": 1, "bf16_nodes": 2}   # include/bsms_hip.h: bsms_precision (bf16: edge MLP; bf16_nodes: edge + node MLP)


def _bsgmp_infer(h, pos, plans, ews, hidden, params, session=None, prec=0):
    B, _, D = h.shape
    p = pos.shape[-1]
    pos_bstride = pos.shape[-2] * p if pos.dim() == 3 else 0
    L = _abi.lib()
    depth = len(plans) - 1
    pl, keep_pl = _abi.ptr_array([q.handle.value if hasattr(q.handle, "value") else q.handle for q in plans])
    ewp, keep_ew = _abi.ptr_array([e.data_ptr() for e in ews])
    out = torch.empty_like(h)
    nbytes = L.bsms_bsgmp_infer_work_bytes(pl, depth, B, D, p, hidden)    # forward-only: without the backward's per-block scratch sets
    reuse = 0
    if session is not None:
        reuse = session.flags(params, plans, B, nbytes, h.device, geom=(D, p, hidden, prec, pos_bstride))
        work = session.work
    else:
        work = _workspace(h.device, nbytes)
    pp, keep = _param_ptrs(params)
    _abi.check(L.bsms_bsgmp_fwd_p(pl, ewp, depth, h.data_ptr(), pos.data_ptr(), B, D, p, pos_bstride, hidden, pp,
                                  out.data_ptr(), None, work.data_ptr(), reuse, prec, _stream()), "bsms_bsgmp_fwd(inference)")
    return out


def _bind_edge_weights(plan, ew):
    """bsms_plan_bind_edge_weights with the bookkeeping its by-ADDRESS binding needs.  Returns the tensor the
    caller should pass as this level's edge weights from now on.

    * The plan object keeps the tensor the C side is bound to alive in `_ew_bound` -- and ONLY that one: the library keeps an
      existing binding to another pointer (captured graphs have the gathered copies baked in), so `_ew_bound` is replaced only
      when the library reports the new pointer as bound.  Dropping the bound tensor would let the caching allocator hand its
      address to another weight tensor of the same size, which the fast path would then take for the bound one.
    * A plan that is already bound to another tensor with the SAME content (a coarse plan whose level-0 plan was evicted from
      the cache and rebuilt: same hierarchy, same weights, new tensors) hands back the bound tensor, so the rebuilt chain
      keeps the compact transition lists instead of falling back to index chasing for good.  One device compare per level
      and rebuilt hierarchy, never on a step's path."""
    L = _abi.lib()
    _abi.check(L.bsms_plan_bind_edge_weights(plan.handle, ew.data_ptr(), _stream()), "bsms_plan_bind_edge_weights")
    bound = L.bsms_plan_bound_edge_weights(plan.handle) or 0
    if bound == ew.data_ptr():
        held = getattr(plan, "_ew_bound", None)
        if held is None or held.data_ptr() != bound:
            plan._ew_bound = ew
        return ew
    held = getattr(plan, "_ew_bound", None)
    if held is not None and held.data_ptr() == bound and held.shape == ew.shape and held.device == ew.device and torch.equal(held, ew):
        return held
    return ew


class BSGMP(nn.Module):
    """ops/BSMS.py:8-104: down pass (GMP, restrict), bottom GMP, up pass (prolong, GMP, skip add).

    Restrict = WeightedEdgeConv + index by m_ids fused (only kept rows are computed); prolong = Unpool +
    WeightedEdgeConv(aggragating=False) fused (zero rows are never materialised)."""

    def __init__(self, unet_depth, latent_dim, hidden_layer, pos_dim):
        super().__init__()
        self.bottom_gmp = GMP(latent_dim, hidden_layer, pos_dim)
        self.down_gmps = nn.ModuleList()
        self.up_gmps = nn.ModuleList()
        self.unpools = nn.ModuleList()
        self.unet_depth = unet_depth
        self.latent_dim, self.pos_dim = latent_dim, pos_dim
        self.per_block = _PY_BSGMP   # True: one autograd node per block / transition (module tree) instead of one call
        # "f32": the reference's arithmetic.  "bf16" (one-call path only): edge-level tensors stored as bf16, bf16 operands
        # in the edge MLP, fp32 accumulation and fp32 everywhere at node level.  "bf16_nodes": the node MLP of every block in
        # that arithmetic as well (include/bsms_hip.h: bsms_precision)
        self.precision = os.environ.get("BSMS_PRECISION", "f32")
        self.edge_conv = WeightedEdgeConv()
        for _ in range(unet_depth):
            self.down_gmps.append(GMP(latent_dim, hidden_layer, pos_dim))
            self.up_gmps.append(GMP(latent_dim, hidden_layer, pos_dim))
            self.unpools.append(Unpool())

    def _edge_weights(self, plans, m_ids, device):
        """cal_ew chain of the down pass (BSMS.py:64,73,89).  It depends on the MESH only (w starts as ones, runs
        under no_grad and never sees h or pos), so it is computed once per hierarchy and cached on the level-0 plan;
        the reference recomputes it in every forward."""
        key = tuple(p.uid for p in plans)
        cache = getattr(plans[0], "_ew_chain", None) if plans else None
        if cache is not None and cache[0] == key:
            return cache[1]
        ews = []
        if plans:
            w = torch.ones(plans[0].N, device=device, dtype=torch.float32)
            for plan, ids in zip(plans, m_ids):
                ew, w_full = self.edge_conv.cal_ew(w, None, plan=plan)
                w = w_full[ids]
                # the weights are mesh-static and cached with the plan: gather them once into the slot orders of the pooled
                # transitions (restrict / prolong then read compact index + weight streams, csrc/rowsum.hip)
                if os.environ.get("BSMS_BIND_EW", "1") == "1":     # ("0": same-box A/B of the unbound index-chasing path, profiles/ab_env.sh)
                    ew = _bind_edge_weights(plan, ew)
                ews.append(ew)
            plans[0]._ew_chain = (key, ews)                      # keyed by plan uids: no reference cycle through the plans
        return ews

    def prepare(self, m_ids, m_gs, n0, device):
        """(plans of levels 0..L-1 with their pools, cached edge weights, plan of the bottom level) of a hierarchy."""
        sizes = [int(n0)] + [int(ids.shape[0]) for ids in m_ids[:self.unet_depth]]      # N_l = number of kept ids of level l - 1
        specs = [(m_gs[i], sizes[i], m_ids[i] if i < self.unet_depth else None) for i in range(self.unet_depth + 1)]
        *plans, bottom = plans_for(specs)                    # new meshes: all levels are built concurrently
        return plans, self._edge_weights(plans, m_ids, device), bottom

    def block_params(self):
        """Parameters in the order of the bsms_bsgmp_* entries: down 0..L-1, bottom, up 0..L-1; node MLP then edge MLP."""
        blocks = [*self.down_gmps, self.bottom_gmp, *self.up_gmps]
        return [q for b in blocks for q in (*b.mlp_node.flat_params(), *b.mlp_edge.flat_params())]

    def forward(self, h, m_ids, m_gs, pos, session=None):
        """`session` (ops.InferenceSession, forward-only calls): reuse weight packs / coarse positions between calls."""
        if h.dim() not in (2, 3) or pos.dim() not in (2, 3):
            raise NotImplementedError("Only implemented for dim 2 and 3")
        h = _dev_f32(h, "BSGMP")
        pos = _dev_f32(pos, "BSGMP")
        L = self.unet_depth
        skips, skip_pos = [], []
        plans, ews, bottom_plan = self.prepare(m_ids, m_gs, h.shape[-2], pos.device)
        n_l = bottom_plan.N
        if plans:
            _check_gmp_shapes("BSGMP", h, pos, plans[0], self.latent_dim, self.pos_dim)
        if not self.per_block:   # the whole U-Net in one library call (csrc/bsgmp.hip)
            squeeze = h.dim() == 2
            if squeeze:
                if pos.dim() == 3:
                    raise NotImplementedError("GMP: 2-D x with 3-D pos is not a layout of the reference")
                h = h.unsqueeze(0)
            all_plans = [*plans, bottom_plan]
            params = self.block_params()
            hidden = self.bottom_gmp.hidden_layer
            prec = PRECISIONS[self.precision]
            if _needs_grad(h, *params):
                y = _BSGMPFunction.apply(h, pos, all_plans, ews, hidden, prec, *params)
            else:
                y = _bsgmp_infer(h, pos, all_plans, ews, hidden, params, session, prec)
            return y.squeeze(0) if squeeze else y
        if self.precision != "f32":
            raise NotImplementedError("the bf16 precision exists on the one-call U-Net path only (BSGMP.per_block = False)")
        for i in range(L):
            plan = plans[i]
            h = self.down_gmps[i](h, m_gs[i], pos, plan=plan)
            skips.append(h)
            skip_pos.append(pos)
            h = _edge_conv(h, ews[i], plan, True, True)          # conv + pool  (BSMS.py:74,79-83)
            with torch.no_grad():
                pos = _edge_conv(pos, ews[i], plan, True, True)  # BSMS.py:75,85-88 ; pos carries no gradient
        h = self.bottom_gmp(h, m_gs[L], pos, plan=plan_for(m_gs[L], h.shape[-2]))
        for i in range(L):
            d = L - 1 - i
            h = _edge_conv(h, ews[d], plans[d], False, True)  # unpool + conv(aggragating=False)  (BSMS.py:98-100)
            h = self.up_gmps[i](h, m_gs[d], skip_pos[d], plan=plans[d])
            h = h + skips[d]
        return h
