"""Mesh-level plans: the Python handle on `bsms_plan_t` (include/bsms_hip.h) plus a cache so that the
reference-style call `GMP()(x, g, pos)` with a raw edge tensor `g` builds its CSR layout once per mesh.

The reference re-derives all indexing from `g` on every call (ops/basic.py:66-72); here `g` [2,E] int64
is converted once into a destination-sorted CSR + source-sorted transpose held in HBM."""
import ctypes as C
import hashlib
import os
import threading
from collections import OrderedDict, deque

import numpy as np
import torch

from . import _abi


_COUNT_LOCK = threading.Lock()
_LOCK = threading.RLock()      # plan cache + interned-index table: a prefetch thread (trainer.DevicePrefetcher) fills them while the training thread reads


def _host_copy(t):
    """CPU int64 copy of an index tensor.  Tensors that came through `intern_index` carry their host original
    (`_bsms_host`): reading a device tensor back would wait for everything queued on the stream -- with variable meshes
    (new plans every batch) that is a pipeline drain per level and step."""
    host = getattr(t, "_bsms_host", None)
    if host is not None and host.shape == t.shape and host.dtype == torch.int64:
        return host
    return t.detach().to("cpu", torch.int64).contiguous()


class LevelPlan:
    """One mesh level (optionally with the kept-node ids of the level for restrict/prolong)."""

    constructed = 0   # number of plans built so far (tests assert that a consistent mesh builds its plans ONCE)

    def __init__(self, g, num_nodes, ids=None, device=None):
        with _COUNT_LOCK:
            LevelPlan.constructed += 1
            self.uid = LevelPlan.constructed      # never reused (unlike id()): cache keys built from it need not keep the plan alive
        g_cpu = _host_copy(g)
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
        self._used = set()
        self.Nk = 0
        self.max_source = int(_abi.lib().bsms_plan_max_source(self._h))
        self.min_out_degree = int(_abi.lib().bsms_plan_min_out_degree(self._h))
        if ids is not None:
            self.set_pool(ids)

    def set_pool(self, ids):
        ids_cpu = np.ascontiguousarray(_host_copy(ids).numpy())
        if self.Nk > 0 and self._used:
            # re-pooling hands the OLD ids / inv block straight back to the library's recycling pool (plan.hip:
            # release_block): kernels of this plan that are still queued read it -- wait for the streams it was used on
            # (plan destruction is ordered by events, _retire; a second set_pool is rare enough for a plain wait)
            for st in list(self._used):
                try:
                    st.synchronize()
                except Exception:
                    pass
        with torch.cuda.device(self.device):
            _abi.check(_abi.lib().bsms_plan_set_pool(self._h, ids_cpu.ctypes.data, int(ids_cpu.shape[0])), "bsms_plan_set_pool")
        self.Nk = int(ids_cpu.shape[0])

    @property
    def handle(self):
        """The `bsms_plan_t*` for a C-ABI call.  Every access notes the current stream: a retired plan is destroyed only
        after the work queued so far on every stream that used it has completed (_retire)."""
        try:
            self._used.add(torch.cuda.current_stream(self.device))
        except Exception:
            pass
        return self._h

    @classmethod
    def from_handle(cls, handle, device):
        """Wrap a `bsms_plan_t*` the library created itself (bsms_plan_concat); ownership passes to the wrapper."""
        self = cls.__new__(cls)
        with _COUNT_LOCK:
            LevelPlan.constructed += 1
            self.uid = LevelPlan.constructed
        L = _abi.lib()
        self.device = torch.device(device)
        self._h = handle
        self._used = set()
        self.N, self.E, self.Nk = int(L.bsms_plan_num_nodes(handle)), int(L.bsms_plan_num_edges(handle)), int(L.bsms_plan_num_pooled(handle))
        self.max_source, self.min_out_degree = int(L.bsms_plan_max_source(handle)), int(L.bsms_plan_min_out_degree(handle))
        return self

    ARRAYS = ("rowptr", "src", "dst", "perm", "t_rowptr", "t_dst", "t_eid", "t_pos", "ids", "inv", "k_rowptr", "k_src", "k_eid",
              "p_rowptr", "p_src", "p_eid", "k_w", "p_w")

    def export_ex(self):
        """Every index array of the plan (include/bsms_hip.h: bsms_plan_export_ex) as {name: int32 numpy array} -- for tests."""
        L, out = _abi.lib(), {}
        for which, name in enumerate(self.ARRAYS):
            n = int(L.bsms_plan_export_ex(self._h, which, None))
            if n < 0:
                raise _abi.BsmsError(f"bsms_plan_export_ex({name})")
            a = np.empty(n, np.int32)
            if n and int(L.bsms_plan_export_ex(self._h, which, a.ctypes.data)) != n:
                raise _abi.BsmsError(f"bsms_plan_export_ex({name})")
            out[name] = a
        return out

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
                _retire(h, getattr(self, "device", None), getattr(self, "_used", ()))
            except Exception:      # interpreter shutdown: module globals are already gone, the process frees the device
                pass


# Plans whose Python handle is gone but whose kernels may still be queued: the library recycles a destroyed plan's device
# block without waiting (include/bsms_hip.h), so a plan is destroyed only after events recorded at retirement on the
# current stream and on every stream the plan was used on have completed (the engine joins its side streams into the
# caller's stream before every return).
_GRAVE = deque()


def _reap():
    """Destroy the retired plans whose events have completed.  `LevelPlan.__del__` runs on whatever thread drops the last
    reference (the training thread, trainer.DevicePrefetcher's thread, a plan-builder thread, or the garbage collector in
    the middle of any allocation) and `bsms_plan_destroy` is a ctypes call that releases the GIL: two threads scanning a
    shared list would both see an entry and destroy its handle twice -- the library would then hand ONE device block to
    two later plans.  Entries are therefore taken OUT of the queue one at a time (`deque.popleft` is atomic: whoever pops
    an entry owns it exclusively), destroyed if finished, and put back if not.  No lock is held at any point, so a
    finaliser that fires inside this function (a nested `_reap`) just pops other entries."""
    for _ in range(len(_GRAVE)):
        try:
            ev, h = _GRAVE.popleft()
        except IndexError:      # another thread got there first
            return
        try:
            done = all(e.query() for e in ev)
        except Exception:
            done = True
        if done:
            try:
                _abi.lib().bsms_plan_destroy(h)
            except Exception:
                pass
        else:
            _GRAVE.append((ev, h))


def _retire(h, device, used=()):
    evs = []
    try:
        if device is not None and torch.cuda.is_available():
            for st in {torch.cuda.current_stream(device), *used}:
                e = torch.cuda.Event()
                e.record(st)
                evs.append(e)
    except Exception:      # interpreter shutdown
        evs = []
    _GRAVE.append((evs, h))
    _reap()


def _cache_capacity():
    """Entries of the plan cache and of the interned-index table.  One mesh of L levels takes L+1 plans and 2L+1 index
    tensors, so the default serves ~70 distinct 7-level meshes before anything is rebuilt (a variable-mesh dataset with
    more should raise BSMS_PLAN_CACHE; an airfoil-size mesh pins ~2 MB of HBM per entry)."""
    return max(int(os.environ.get("BSMS_PLAN_CACHE", "1024")), 8)


def _key(t):
    st = t.untyped_storage()
    return (st.data_ptr(), t.storage_offset(), tuple(t.shape), tuple(t.stride()), t._version, str(t.device))


class _PlanCache:
    """LRU keyed by the identity of the index tensors' storage (+ version counter).  Each entry keeps
    the storages alive, so a cached address can never be recycled for a different graph."""

    def __init__(self, capacity=None):
        self.capacity = _cache_capacity() if capacity is None else capacity
        self._d = OrderedDict()

    def get(self, g, num_nodes, ids=None):
        key = (_key(g), int(num_nodes), _key(ids) if ids is not None else None)
        with _LOCK:
            hit = self._d.get(key)
            if hit is not None:
                self._d.move_to_end(key)
                return hit[0]
            if _GRAVE:
                _reap()
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


_BUILDERS = None


def plans_for(specs):
    """plan_for over a list of (g, num_nodes, ids): the plans that are not cached are built CONCURRENTLY (the host CSR
    build of bsms_plan_create runs without the GIL; a variable-mesh batch needs L + 1 new plans every step, 2.3 ms
    serially at cylinder size -- about the whole GPU step)."""
    global _BUILDERS
    keys = [(_key(g), int(n), _key(ids) if ids is not None else None) for g, n, ids in specs]
    with _LOCK:
        missing = [i for i, k in enumerate(keys) if k not in _CACHE._d]
        if len(missing) >= 2:
            if _BUILDERS is None:
                from concurrent.futures import ThreadPoolExecutor
                _BUILDERS = ThreadPoolExecutor(max_workers=8, thread_name_prefix="bsms-plan")
            if _GRAVE:
                _reap()
    if len(missing) >= 2:                                  # built WITHOUT the lock: the training thread's lookups go on meanwhile
        built = list(_BUILDERS.map(lambda i: LevelPlan(*specs[i]), missing))
        with _LOCK:
            for i, plan in zip(missing, built):
                if keys[i] not in _CACHE._d:               # (another thread may have built the same plan: the first one stays)
                    g, _, ids = specs[i]
                    _CACHE._d[keys[i]] = (plan, g.untyped_storage(), ids.untyped_storage() if ids is not None else None)
            while len(_CACHE._d) > _CACHE.capacity:
                _CACHE._d.popitem(last=False)
    return [_CACHE.get(g, n, ids) for g, n, ids in specs]


def clear_plan_cache():
    _CACHE.clear()
    _INTERNED.clear()


# ------------------------------------------------------------------------------------ interned index tensors
# A training loop that follows the reference (trainer/trainer.py:143 `move_to_device` on every batch) hands the model
# FRESH device copies of the same edge lists every step; a cache keyed on tensor identity then misses every step and
# every miss costs a device->host copy, a host CSR build and ~10 synchronous allocations.  Index tensors are therefore
# interned by CONTENT while they are still on the host: equal content -> the same device tensor object -> the identity
# keyed plan cache hits, and the edge lists are not re-uploaded either.
_INTERNED = OrderedDict()
try:                                   # optional accelerator; the stdlib digest below is the fallback
    import xxhash as _xxhash
except ImportError:
    _xxhash = None


def _content_key(t):
    a = np.ascontiguousarray(t.numpy()).view(np.uint8).reshape(-1)
    digest = _xxhash.xxh3_128_intdigest(a) if _xxhash is not None else hashlib.blake2b(a, digest_size=16).digest()
    return (tuple(t.shape), str(t.dtype), digest)


_COPY_STREAMS = {}


def _upload(t, device):
    """Host -> device without waiting for the stream: a blocking `.to()` of pageable memory returns only after everything
    queued before it has run (a pipeline drain per tensor when every batch brings new tensors).  Pinned staging comes
    from PyTorch's caching host allocator, which keeps the block alive until the copy has executed."""
    dev = torch.device(device)
    if not t.is_cuda and dev.type == "cuda":
        # on a copy stream of its own: a DMA copy queued between two kernels of the compute stream costs ~0.1 ms of queue
        # switching each (three per step = the whole difference between a host-fed and a device-resident step), while
        # the host runs a step ahead of the GPU, so on its own stream the copy has long finished when the step starts
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        cur = torch.cuda.current_stream(idx)
        cs = _COPY_STREAMS.get(idx)
        if cs is None:
            # a stream that really runs beside the compute stream: HIP maps streams onto four hardware queues and two streams on
            # one queue run in order -- take the first of torch's pool streams that the probe says can overtake `cur`
            tried = []
            for _ in range(6):
                cs = torch.cuda.Stream(idx)
                tried.append(cs)
                try:
                    if os.environ.get("BSMS_COPY_STREAM_PROBE", "1") == "0" or _abi.lib().bsms_streams_overlap(cur.cuda_stream, cs.cuda_stream) == 1:
                        break
                except Exception:      # noqa: BLE001  (no library / no GPU: any stream will do)
                    break
            _COPY_STREAMS[idx] = cs
        src = t if t.is_pinned() else t.pin_memory()
        with torch.cuda.stream(cs):
            out = src.to(torch.device("cuda", idx), non_blocking=True)
        cur.wait_stream(cs)
        out.record_stream(cur)
        return out
    return t.to(device)


def _interned(key, src):
    """The interned device tensor of `key` if its host original equals `src` (a digest match is confirmed by memcmp)."""
    with _LOCK:
        hit = _INTERNED.get(key)
        if hit is not None:
            _INTERNED.move_to_end(key)
    return hit[0] if hit is not None and torch.equal(hit[1], src) else None


def intern_index(t, device, shared_batch_axis=False):
    """Device copy of the CPU int64 index tensor `t`, shared between calls with equal content.
    shared_batch_axis: `t` is [B, ...] and the consumer reads t[0] only (consistent-mesh collate, models/model.py:190-192);
    when all B slices are equal only ONE slice is uploaded and the result is an expanded (stride-0) view, so the cache
    does not pin B copies of the edge list in HBM.
    (The table is shared with trainer.DevicePrefetcher's thread: hashing, comparing and uploading run outside its lock.)"""
    if t.is_cuda or t.dtype != torch.int64:
        return t.to(device)
    first = None
    if shared_batch_axis and t.dim() >= 2 and t.shape[0] >= 1:
        # hot path (every step of a consistent-mesh loader): digest + memcmp of slice 0 only -- 1/B of the bytes; the
        # other slices were compared with slice 0 when this content was first seen, and the consumer reads t[0] only
        first = t[0]
        hit = _interned((_content_key(first), t.shape[0], str(device)), first)
        if hit is not None:
            return hit
        if not bool((t == t[:1]).all()):                   # slices differ: not a shared axis after all
            first = None
    key = (_content_key(first if first is not None else t), t.shape[0] if first is not None else -1, str(device))
    src = first if first is not None else t
    hit = _interned(key, src)
    if hit is not None:
        return hit
    host = src.contiguous().clone()
    if first is not None:
        dev = _upload(host, device).unsqueeze(0).expand(t.shape[0], *first.shape)
    else:
        dev = _upload(host, device)
        dev._bsms_host = host                  # LevelPlan builds its CSR from the host original (no device read-back)
    with _LOCK:
        _INTERNED[key] = (dev, host)
        if len(_INTERNED) > _cache_capacity():
            _INTERNED.popitem(last=False)
    return dev


class LevelData:
    """Stand-in for the per-level `torch_geometric.data.Data` the reference datapipe yields for variable
    meshes (datasets/base.py:325-349): level 0 carries x / y / mask, every level carries `edge_index`, and the
    kept-node ids ride in `face` (the reference abuses `face` so that PyG offsets them like an index)."""

    def __init__(self, edge_index, num_nodes, face=None, x=None, y=None, mask=None):
        self.edge_index, self.num_nodes, self.face, self.x, self.y, self.mask = edge_index, num_nodes, face, x, y, mask

    def to(self, device, intern=False):
        mv = lambda t: None if t is None else _upload(t, device)
        mi = (lambda t: None if t is None else intern_index(t, device)) if intern else mv
        return LevelData(mi(self.edge_index), self.num_nodes, mi(self.face), mv(self.x), mv(self.y), mv(self.mask))


def collate_variable_meshes(samples):
    """What `torch_geometric.loader.DataLoader` (PyG 2.5.3 `Batch`, train.py:50) does to a list of samples, each a
    list of per-level LevelData: node tensors are concatenated on dim 0, `edge_index` and `face` on the last dim
    with the CUMULATIVE NUMBER OF NODES OF THAT LEVEL added.  Result: one block-diagonal graph per level, which
    `BSMS_Simulator.forward(data, consistent_mesh=False, ...)` consumes with a batch axis of 1."""
    out = []
    for lvl in range(len(samples[0])):
        parts = [s[lvl] for s in samples]
        offs, acc = [], 0
        for d in parts:
            offs.append(acc)
            acc += int(d.num_nodes)
        cat = lambda ts, dim: None if ts[0] is None else torch.cat(ts, dim)
        out.append(LevelData(
            edge_index=torch.cat([d.edge_index + o for d, o in zip(parts, offs)], dim=-1), num_nodes=acc,
            face=None if parts[0].face is None else torch.cat([d.face + o for d, o in zip(parts, offs)], dim=-1),
            x=cat([d.x for d in parts], 0), y=cat([d.y for d in parts], 0), mask=cat([d.mask for d in parts], 0)))
    return out


# ------------------------------------------------------------------------------------ variable meshes, collated on the device
def concat_plans(parts, ew_cat=None, want_index=True):
    """Block-diagonal union of the LevelPlans `parts` built on the GPU (bsms_plan_concat): (plan, edge_index [2, E] int64, kept ids
    [Nk] int64 or None) -- what `collate_variable_meshes` + `LevelPlan(...)` would give for the same meshes, without a host CSR
    build or an upload.  `ew_cat`: concatenation of the parts' BOUND edge-weight tensors (the union is then bound to it).
    Stream-ordered on the current stream."""
    from .ops import _stream
    L = _abi.lib()
    dev = parts[0].device
    E, Nk = sum(q.E for q in parts), sum(q.Nk for q in parts)
    pooled = parts[0].Nk > 0
    coo = torch.empty(2, E, dtype=torch.int64, device=dev) if want_index else None
    ids = torch.empty(Nk, dtype=torch.int64, device=dev) if (want_index and pooled) else None
    pl, keep = _abi.ptr_array([q.handle.value if hasattr(q.handle, "value") else q.handle for q in parts])
    out = C.c_void_p()
    with torch.cuda.device(dev):
        _abi.check(L.bsms_plan_concat(pl, len(parts), None if ew_cat is None else ew_cat.data_ptr(),
                                      None if coo is None else coo.data_ptr(), None if ids is None else ids.data_ptr(),
                                      _stream(), C.byref(out)), "bsms_plan_concat")
    plan = LevelPlan.from_handle(out, dev)
    plan._parts = tuple(parts)          # the union copied the parts' arrays: nothing is shared, but a reader may want to know
    if ew_cat is not None:
        plan._ew_bound = ew_cat         # bound by ADDRESS: the plan keeps the tensor alive (ops._bind_edge_weights)
    return plan, coo, ids


class MeshBank:
    """Per-MESH device state for variable-mesh training (the reference's cylinder_flow path, `consistent_mesh: false`: every
    batch is a new combination of meshes, datasets/base.py:319-351).  A mesh's hierarchy is uploaded ONCE, its plans (CSR,
    transpose, pooled transitions) and its edge-weight chain are built ONCE and stay in HBM; `collate(samples)` then assembles a
    batch on the GPU: one `bsms_plan_concat` per level + one concatenation of the cached edge weights per level -- no host
    collate of index lists, no index upload, no CSR build.  The result is what `collate_variable_meshes` + `.to(device)` give
    (same tensors, PyG `Batch` semantics), with the plans and the edge-weight chain of the batch already in the engine's caches.

    `process`: the model's BSGMP (it owns `prepare`, the per-hierarchy plan / edge-weight construction)."""

    def __init__(self, process, device, capacity=4096):
        self.process, self.device, self.capacity = process, torch.device(device), capacity
        self._by_obj, self._by_content = OrderedDict(), OrderedDict()

    def entry(self, levels):
        """The resident state of ONE mesh, given its per-level LevelData (host): looked up by the identity of its level-0 edge
        tensor first (a dataset that keeps its meshes in memory hands out the same objects), by content otherwise."""
        k = id(levels[0].edge_index)
        hit = self._by_obj.get(k)
        if hit is not None and hit[0] is levels[0].edge_index:
            return hit[1]
        dev_idx = [(intern_index(d.edge_index, self.device), None if d.face is None else intern_index(d.face, self.device)) for d in levels]
        ck = tuple((id(g), None if f is None else id(f)) for g, f in dev_idx)      # interned: equal content -> the same device tensors
        ent = self._by_content.get(ck)
        if ent is None:
            depth = len(levels) - 1
            m_gs, m_ids = [g for g, _ in dev_idx], [f for _, f in dev_idx[:depth]]
            plans, ews, bottom = self.process.prepare(m_ids, m_gs, int(levels[0].num_nodes), self.device)
            ent = dict(plans=[*plans, bottom], ews=list(ews), keep=dev_idx)
            self._by_content[ck] = ent
            while len(self._by_content) > self.capacity:
                self._by_content.popitem(last=False)
        self._by_obj[k] = (levels[0].edge_index, ent)
        while len(self._by_obj) > self.capacity:
            self._by_obj.popitem(last=False)
        return ent

    def collate(self, samples):
        """`samples`: list of per-sample lists of host LevelData (what `collate_variable_meshes` takes).  Returns the per-level
        device LevelData of the block-diagonal batch."""
        ents = [self.entry(s) for s in samples]
        depth = len(samples[0]) - 1
        cat0 = lambda get: None if get(samples[0][0]) is None else _upload(torch.cat([get(s[0]) for s in samples], 0), self.device)
        x, y, mask = cat0(lambda d: d.x), cat0(lambda d: d.y), cat0(lambda d: d.mask)
        out, plans, ews = [], [], []
        for lvl in range(depth + 1):
            parts = [e["plans"][lvl] for e in ents]
            ew_cat = torch.cat([e["ews"][lvl] for e in ents]) if lvl < depth else None
            # the union may take the parts' gathered weight copies only if every part is bound to ITS OWN chain tensor (a coarse plan
            # shared by two hierarchies keeps the first binding, ops._bind_edge_weights): otherwise the union stays unbound
            bound = ew_cat is not None and all((_abi.lib().bsms_plan_bound_edge_weights(q.handle) or 0) == e["ews"][lvl].data_ptr()
                                               for q, e in zip(parts, ents))
            plan, coo, ids = concat_plans(parts, ew_cat if bound else None)
            key = (_key(coo), plan.N, _key(ids) if ids is not None else None)
            with _LOCK:                                   # the engine's lookup by tensor identity (plan_for / plans_for) hits
                _CACHE._d[key] = (plan, coo.untyped_storage(), ids.untyped_storage() if ids is not None else None)
                while len(_CACHE._d) > _CACHE.capacity:
                    _CACHE._d.popitem(last=False)
            plans.append(plan)
            if ew_cat is not None:
                ews.append(ew_cat)
            out.append(LevelData(edge_index=coo, num_nodes=plan.N, face=ids, x=x if lvl == 0 else None, y=y if lvl == 0 else None,
                                 mask=mask if lvl == 0 else None))
        if depth > 0:
            plans[0]._ew_chain = (tuple(q.uid for q in plans[:depth]), ews)   # BSGMP._edge_weights: the chain of this hierarchy, ready
        return out
