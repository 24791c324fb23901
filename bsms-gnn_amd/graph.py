"""Mesh-level plans: the Python handle on `bsms_plan_t` (include/bsms_hip.h) plus a cache so that the
reference-style call `GMP()(x, g, pos)` with a raw edge tensor `g` builds its CSR layout once per mesh.

The reference re-derives all indexing from `g` on every call (ops/basic.py:66-72); here `g` [2,E] int64
is converted once into a destination-sorted CSR + source-sorted transpose held in HBM."""
import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _abi


class LevelPlan:
    """One mesh level (optionally with the kept-node ids of the level for restrict/prolong)."""

    def __init__(self, g, num_nodes, ids=None, device=None):
        g_cpu = g.detach().to("cpu", torch.int64).contiguous()
        if g_cpu.dim() != 2 or g_cpu.shape[0] != 2:
            raise ValueError(f"edge list must be [2, E], got {tuple(g_cpu.shape)}")
        self.device = torch.device(device if device is not None else g.device)
        if self.device.type != "cuda":
            raise _abi.BsmsError("LevelPlan needs a GPU device: the BSMS engine has no CPU path")
        self.E, self.N = int(g_cpu.shape[1]), int(num_nodes)
        coo = np.ascontiguousarray(g_cpu.numpy())
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _abi.check(_abi.lib().bsms_plan_create(coo.ctypes.data, self.E, self.N, C.byref(handle)), "bsms_plan_create")
        self._h = handle
        self.Nk = 0
        self.max_source = int(_abi.lib().bsms_plan_max_source(self._h))
        self.min_out_degree = int(_abi.lib().bsms_plan_min_out_degree(self._h))
        if ids is not None:
            self.set_pool(ids)

    def set_pool(self, ids):
        ids_cpu = np.ascontiguousarray(ids.detach().to("cpu", torch.int64).numpy())
        with torch.cuda.device(self.device):
            _abi.check(_abi.lib().bsms_plan_set_pool(self._h, ids_cpu.ctypes.data, int(ids_cpu.shape[0])), "bsms_plan_set_pool")
        self.Nk = int(ids_cpu.shape[0])

    @property
    def handle(self):
        return self._h

    def export(self):
        """(rowptr, src_sorted, perm, t_rowptr) as int32 numpy arrays -- for tests."""
        rp = np.empty(self.N + 1, np.int32)
        src = np.empty(self.E, np.int32)
        perm = np.empty(self.E, np.int32)
        trp = np.empty(self.N + 1, np.int32)
        _abi.check(_abi.lib().bsms_plan_export(self._h, rp.ctypes.data, src.ctypes.data, perm.ctypes.data, trp.ctypes.data),
                   "bsms_plan_export")
        return rp, src, perm, trp

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _abi.lib().bsms_plan_destroy(h)
            except Exception:
                pass


def _key(t):
    st = t.untyped_storage()
    return (st.data_ptr(), t.storage_offset(), tuple(t.shape), tuple(t.stride()), t._version, str(t.device))


class _PlanCache:
    """LRU keyed by the identity of the index tensors' storage (+ version counter).  Each entry keeps
    the storages alive, so a cached address can never be recycled for a different graph."""

    def __init__(self, capacity=256):
        self.capacity = capacity
        self._d = OrderedDict()

    def get(self, g, num_nodes, ids=None):
        key = (_key(g), int(num_nodes), _key(ids) if ids is not None else None)
        hit = self._d.get(key)
        if hit is not None:
            self._d.move_to_end(key)
            return hit[0]
        plan = LevelPlan(g, num_nodes, ids)
        self._d[key] = (plan, g.untyped_storage(), ids.untyped_storage() if ids is not None else None)
        if len(self._d) > self.capacity:
            self._d.popitem(last=False)
        return plan

    def clear(self):
        self._d.clear()


_CACHE = _PlanCache()


def plan_for(g, num_nodes, ids=None):
    """Cached LevelPlan for edge tensor `g` ([2,E] int64 on the GPU) of a level with `num_nodes`."""
    return _CACHE.get(g, num_nodes, ids)


def clear_plan_cache():
    _CACHE.clear()
