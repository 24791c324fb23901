"""One training step of `BSMS_Simulator` -- forward, masked-RMSE loss, backward -- as a fixed sequence of C-ABI calls on
static buffers, without autograd (reference: models/model.py:127-164, trainer/trainer.py:79-98,144-147).

Why: the autograd mirror of the reference (model.py / ops.py) spends ~40 tiny element-wise launches per step on the
normaliser and the loss, re-creates ~190 gradient views and pointer tables in Python every backward, and cannot be
captured in a HIP graph.  Here the step is

    bsms_sim_prologue -> bsms_mlp_fwd (encode) -> bsms_bsgmp_fwd (process) -> bsms_mlp_fwd (decode) -> bsms_sim_epilogue
    [data parallel: all-reduce of the two loss sums]
    bsms_sim_loss_bwd -> bsms_mlp_bwd (decode) -> bsms_bsgmp_bwd (process) -> bsms_mlp_bwd (encode)
    [data parallel: all-reduce of the flat gradient buffer]

with every weight gradient written straight into its slot of the flat gradient buffer (`GradBuckets.flat`), which the
parameters' `.grad` alias.  Same kernels as the autograd path for everything but the glue: U-Net / MLP gradients are
bit-identical to it, the loss and its gradient agree to fp32 round-off (tests/test_hip_training.py).  With
`use_graph=True` the two halves are captured into HIP graphs and replayed (inputs are copied into static buffers)."""
import os

import torch
import torch.distributed as dist

from . import _abi
from .ops import PRECISIONS, _param_ptrs, _stream


def _ptr(t):
    return None if t is None else t.data_ptr()


class _Arena:
    """Grow-only device buffers by name.  With variable meshes (the reference's cylinder_flow path) every batch has its
    own node / edge counts: buffers sized exactly would be released and re-allocated every step -- hundreds of MB through
    hipMalloc, 30-70 ms per step against a 2.5 ms step (profiles/fresh_mesh.py).  A buffer is re-allocated only when
    a batch needs more than its capacity, then with 25 % headroom."""

    def __init__(self):
        self._t = {}

    def bytes(self, name, nbytes, dev):
        nbytes = max(int(nbytes), 1)
        t = self._t.get(name)
        if t is None or t.numel() < nbytes or t.device != dev:
            cap = nbytes if t is None else nbytes + nbytes // 4
            self._t[name] = None                     # release the old block before asking for the larger one
            t = self._t[name] = torch.empty((cap + 255) // 256 * 256, device=dev, dtype=torch.uint8)
        return t[:nbytes]

    def f32(self, name, dev, *shape):
        n = 1
        for d in shape:
            n *= int(d)
        return self.bytes(name, 4 * max(n, 1), dev)[:4 * n].view(torch.float32).view(*shape)


class FusedStep:
    def __init__(self, model, grads, group=None, use_graph=False):
        from .model import BSMS_Simulator
        if not isinstance(model, BSMS_Simulator):
            raise TypeError("FusedStep drives a bsms_gnn_amd.BSMS_Simulator")
        if model.process.per_block:
            raise ValueError("FusedStep uses the one-call U-Net (BSGMP.per_block must be False)")
        self.model, self.grads, self.group, self.use_graph = model, grads, group, use_graph
        self._shape_key, self._graphs, self._ptr_guard = None, None, None
        self._arena = _Arena()
        self._overlap = None          # bucket schedule of the overlapped gradient all-reduce (_bucket_schedule), built lazily
        self._comm = None             # communication stream the bucket all-reduces are issued from
        for p in grads.params:                      # .grad aliases the flat buffer once and for all
            off, n = grads._slot[p]
            p.grad = grads.flat[off:off + n].view_as(p)

    # ------------------------------------------------------------------------------------------------ helpers
    @staticmethod
    def supports(model):
        from .model import BSMS_Simulator
        return (isinstance(model, BSMS_Simulator) and not model.process.per_block
                and all(p.requires_grad for p in [*model.encode.parameters(), *model.process.parameters(), *model.decode.parameters()]))

    collectives_at_world_one = False     # tests: issue the step's collectives in a process group of ONE rank as well (a sum over one
                                         # rank is the identity) -- the only way to run the RCCL path on a single-GPU box

    def _world(self):
        w = dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1
        return 2 if (w == 1 and self.collectives_at_world_one and dist.is_available() and dist.is_initialized()) else w

    def _unpack(self, data, consistent):
        if consistent:
            node_in, tar, mask, m_gs, m_ids = data
            m_gs, m_ids = [g[0] for g in m_gs], [i[0] for i in m_ids]
        else:
            node_in, tar, mask = data[0].x.unsqueeze(0), data[0].y.unsqueeze(0), data[0].mask.unsqueeze(0)
            m_gs = [d.edge_index for d in data]
            m_ids = [data[i].face for i in range(len(m_gs) - 1)]
        self._validate(node_in, tar, mask, m_gs, m_ids)
        f = lambda t: t if (t.is_contiguous() and t.dtype == torch.float32) else t.contiguous().float()
        return f(node_in), f(tar), f(mask), m_gs, m_ids

    def _validate(self, node_in, tar, mask, m_gs, m_ids):
        """The C entries take raw pointers with sizes from cfg: a mis-shaped batch must fail HERE, like the shape
        error PyTorch would raise in the reference (models/model.py:127-164), not read device memory out of bounds."""
        cfg = self.model.cfg
        C, p, depth = cfg.out_dim, self.model.pos_dim, cfg.unet_depth
        if node_in.dim() != 3 or node_in.shape[-1] != C + p + 1:
            raise RuntimeError(f"node_in must be [B, N, out_dim + pos_dim + 1 = {C + p + 1}], got {tuple(node_in.shape)}")
        B, N = node_in.shape[0], node_in.shape[1]
        if tuple(tar.shape) != (B, N, C):
            raise RuntimeError(f"target must be [B, N, out_dim] = {(B, N, C)}, got {tuple(tar.shape)}")
        if mask.numel() != B * N or (mask.dim() == 3 and tuple(mask.shape) != (B, N, 1)) or mask.dim() not in (2, 3):
            raise RuntimeError(f"mask must be [B, N, 1] = {(B, N, 1)}, got {tuple(mask.shape)}")
        if len(m_gs) != depth + 1 or len(m_ids) != depth:
            raise RuntimeError(f"unet_depth = {depth} needs {depth + 1} edge lists and {depth} kept-id lists, "
                               f"got {len(m_gs)} and {len(m_ids)}")
        dev = node_in.device
        for name, t in (("target", tar), ("mask", mask), *((f"m_gs[{i}]", g) for i, g in enumerate(m_gs)),
                        *((f"m_ids[{i}]", g) for i, g in enumerate(m_ids))):
            if t.device != dev:
                raise RuntimeError(f"{name} is on {t.device}, node_in on {dev}: all tensors of a step must share one device")
        if next(self.model.parameters()).device != dev:
            raise RuntimeError(f"the model is on {next(self.model.parameters()).device}, the batch on {dev}")

    def _pointer_tables(self):
        m = self.model
        enc, proc, dec = m.encode.flat_params(), m.process.block_params(), m.decode.flat_params()
        guard = (enc[0].data_ptr(), proc[-1].data_ptr(), dec[-1].data_ptr())
        if guard != self._ptr_guard:                # parameters were re-pointed (optimizer flat buffer, .to(), load)
            slot = lambda ps: [self.grads.flat[self.grads._slot[p][0]:].data_ptr() for p in ps]
            self._tabs = {k: (_param_ptrs(ps), _abi.ptr_array(slot(ps))) for k, ps in (("enc", enc), ("proc", proc), ("dec", dec))}
            self._ptr_guard = guard
            self._graphs = None
        return self._tabs

    def _buffers(self, B, N, plans, dev):
        m, L = self.model, _abi.lib()
        C, p, D, H = m.cfg.out_dim, m.pos_dim, m.cfg.latent_dim, m.cfg.hidden_layer
        key = (B, N, tuple(q.uid for q in plans), str(dev), m.process.precision)
        if key == self._shape_key:
            return self._buf
        R, depth = B * N, len(plans) - 1
        pl, keep = _abi.ptr_array([q.handle.value if hasattr(q.handle, "value") else q.handle for q in plans])
        ar = self._arena                              # views of grow-only buffers: a batch of another size re-uses the memory
        f = lambda name, *s: ar.f32(name, dev, *s)
        u8 = lambda name, n: ar.bytes(name, n, dev)
        b = dict(R=R, pl=pl, pl_keep=keep, plans=plans, depth=depth,
                 norm_in=f("norm_in", R, C + 1), pos=f("pos", R, p), h0=f("h0", R, D), h1=f("h1", R, D),
                 norm_pred=f("norm_pred", R, C), pred=f("pred", B, N, C),
                 sums=f("sums", 2), loss=f("loss", 1), g_np=f("g_np", R, C), gh1=f("gh1", R, D), gh0=f("gh0", R, D),
                 s_enc=u8("s_enc", L.bsms_mlp_saved_bytes(R, C + 1, D, D, H)), s_dec=u8("s_dec", L.bsms_mlp_saved_bytes(R, D, D, C, H)),
                 s_proc=u8("s_proc", L.bsms_bsgmp_saved_bytes_p(pl, depth, B, D, p, H, PRECISIONS[m.process.precision])),
                 prec=m.process.precision,
                 work=u8("work", max(L.bsms_mlp_work_bytes(R, C + 1, D, D, H), L.bsms_mlp_work_bytes(R, D, D, C, H),
                                     L.bsms_bsgmp_work_bytes(pl, depth, B, D, p, H), L.bsms_sim_work_bytes(R))),
                 work_enc=u8("work_enc", L.bsms_mlp_work_bytes(R, C + 1, D, D, H)),   # the encoder's backward overlaps the U-Net's last weight gradients
                 work_dec=u8("work_dec", L.bsms_mlp_work_bytes(R, D, D, C, H)),       # the decoder's weight gradients run under the U-Net's first block
                 in_static=None)
        self._shape_key, self._buf, self._graphs = key, b, None
        return b

    # ------------------------------------------------------------------------------------------------ the two halves
    def _forward(self, b, node_in, tar, mask, ews, B, N):
        m, L, s = self.model, _abi.lib(), _stream()
        C, p, D, H = m.cfg.out_dim, m.pos_dim, m.cfg.latent_dim, m.cfg.hidden_layer
        R, t = b["R"], self._tabs
        work = b["work"]                      # owned scratch: the launches are ordered on one stream, also under capture
        ni, no = m._inputNormalizer, m._targetNormalizer
        ck = _abi.check
        ck(L.bsms_sim_prologue(node_in.data_ptr(), R, C, p, ni._E_data.data_ptr(), ni._E_data_squared.data_ptr(),
                               ni.std_eps.data_ptr(), b["norm_in"].data_ptr(), b["pos"].data_ptr(), s), "bsms_sim_prologue")
        ck(L.bsms_mlp_fwd(b["norm_in"].data_ptr(), R, C + 1, D, D, H, 1, t["enc"][0][0], b["h0"].data_ptr(), b["s_enc"].data_ptr(),
                          work.data_ptr(), s), "bsms_mlp_fwd(encode)")
        ewp, keep = _abi.ptr_array([e.data_ptr() for e in ews])
        ck(L.bsms_bsgmp_fwd_p(b["pl"], ewp, b["depth"], b["h0"].data_ptr(), b["pos"].data_ptr(), B, D, p, N * p, H, t["proc"][0][0],
                              b["h1"].data_ptr(), b["s_proc"].data_ptr(), work.data_ptr(), 0, PRECISIONS[b["prec"]], s), "bsms_bsgmp_fwd")
        ck(L.bsms_mlp_fwd(b["h1"].data_ptr(), R, D, D, C, H, 0, t["dec"][0][0], b["norm_pred"].data_ptr(), b["s_dec"].data_ptr(),
                          work.data_ptr(), s), "bsms_mlp_fwd(decode)")
        ck(L.bsms_sim_epilogue(b["norm_pred"].data_ptr(), node_in.data_ptr(), mask.data_ptr(), tar.data_ptr(), R, C, p,
                               no._E_data.data_ptr(), no._E_data_squared.data_ptr(), no.std_eps.data_ptr(), b["pred"].data_ptr(),
                               None, None, b["sums"].data_ptr(), work.data_ptr(), s), "bsms_sim_epilogue")

    def _backward(self, b, tar, mask, ews, B, N, events=None):
        """`events`: (pointer array, keep-alive) of 2L+1 hipEvent_t for bsms_bsgmp_bwd_ev, or None."""
        m, L, s = self.model, _abi.lib(), _stream()
        C, p, D, H = m.cfg.out_dim, m.pos_dim, m.cfg.latent_dim, m.cfg.hidden_layer
        R, t = b["R"], self._tabs
        work = b["work"]
        no = m._targetNormalizer
        ck = _abi.check
        ck(L.bsms_sim_loss_bwd(b["pred"].data_ptr(), tar.data_ptr(), mask.data_ptr(), R, C, no._E_data.data_ptr(),
                               no._E_data_squared.data_ptr(), no.std_eps.data_ptr(), b["sums"].data_ptr(), b["loss"].data_ptr(),
                               b["g_np"].data_ptr(), s), "bsms_sim_loss_bwd")
        ck(L.bsms_mlp_bwd_ex(b["h1"].data_ptr(), b["g_np"].data_ptr(), R, D, D, C, H, 0, t["dec"][0][0], b["s_dec"].data_ptr(),
                             b["work_dec"].data_ptr(), b["gh1"].data_ptr(), t["dec"][1][0], 1, s), "bsms_mlp_bwd(decode)")   # 1 = BSMS_BWD_DEFER_JOIN
        ewp, keep = _abi.ptr_array([e.data_ptr() for e in ews])
        # BSMS_BWD_DEFER_JOIN: the weight gradients of the last (level-0) block are still running on the engine's side
        # streams (~0.2 ms on half the chip) while the encoder's backward -- own scratch, own gradient slots -- runs here
        ck(L.bsms_bsgmp_bwd_ev(b["pl"], ewp, b["depth"], b["h0"].data_ptr(), b["pos"].data_ptr(), b["gh1"].data_ptr(), B, D, p, N * p, H,
                               t["proc"][0][0], b["s_proc"].data_ptr(), work.data_ptr(), b["gh0"].data_ptr(), t["proc"][1][0],
                               PRECISIONS[b["prec"]], 1, None if events is None else events[0], s), "bsms_bsgmp_bwd")
        ck(L.bsms_mlp_bwd(b["norm_in"].data_ptr(), b["gh0"].data_ptr(), R, C + 1, D, D, H, 1, t["enc"][0][0], b["s_enc"].data_ptr(),
                          b["work_enc"].data_ptr(), None, t["enc"][1][0], s), "bsms_mlp_bwd(encode)")
        ck(L.bsms_side_lanes_join(s), "bsms_side_lanes_join")

    # ------------------------------------------------------------------------------------------------ the step
    def __call__(self, data, consistent=True):
        node_in, tar, mask, m_gs, m_ids = self._unpack(data, consistent)
        if not node_in.is_cuda:
            raise _abi.BsmsError("FusedStep: the BSMS engine runs on the GPU only; there is no CPU fallback")
        B, N = node_in.shape[0], node_in.shape[1]
        plans, ews, bottom = self.model.process.prepare(m_ids, m_gs, N, node_in.device)
        plans = [*plans, bottom]
        self._pointer_tables()
        b = self._buffers(B, N, plans, node_in.device)
        world = self._world()
        if self.use_graph:
            return self._replay(b, node_in, tar, mask, ews, B, N, world)
        self._forward(b, node_in, tar, mask, ews, B, N)
        if world > 1:
            dist.all_reduce(b["sums"], op=dist.ReduceOp.SUM, group=self.group)
        if world > 1 and self._overlap_now():
            self._backward_overlapped(b, tar, mask, ews, B, N)
        else:
            self._backward(b, tar, mask, ews, B, N)
            if world > 1:
                dist.all_reduce(self.grads.flat, op=dist.ReduceOp.SUM, group=self.group)
        self._probe_end()
        return b["loss"][0].clone()       # the static buffer is overwritten by the next step: hand out a copy (4 bytes)

    # ------------------------------------------------------------------------------------------------ overlapped all-reduce
    # False (BSMS_OVERLAP_ALLREDUCE=0): ONE all-reduce of the whole flat buffer after the backward (rounds 1-3)
    overlap_allreduce = os.environ.get("BSMS_OVERLAP_ALLREDUCE", "1") == "1"

    overlap_min_gain = 0.97       # the overlapped form is kept only if its step time <= this x the plain form's (>= 3 % gain)
    force_overlap = False         # tests: take the overlapped path on any backend, without the self-check below
    probe_any_backend = False     # tests: run the self-check on a backend other than nccl

    def _overlap_now(self):
        """Overlapped (per-bucket) or plain (one message) gradient all-reduce for THIS step.
        * Only on the nccl backend (= RCCL): there an asynchronous collective issued from a side stream is stream-ordered
          and costs the host nothing.  gloo stages CUDA tensors through the host from a worker thread; four asynchronous
          works per step behind stream-side event waits measured 0.1-2 s per step against 8 ms for one blocking message
          (profiles/dbg_overlap.py) -- the CPU tests and the one-GPU functional runs use the plain form.
        * SELF-CHECK: no multi-GPU machine was available to the builder, so the first four data-parallel steps of a process
          measure both forms (two plain, two overlapped, each timed with HIP events and synchronised), the ranks agree on
          the maxima with one tiny all-reduce, and the overlapped form is kept only if it PROVES a gain of at least 3 %
          (round 5; round 4 kept it unless it was 1.25x slower -- the plain single message is the de-risked default, the
          overlap has to earn its machinery).  The decision is identical on every rank (a mixed choice would mismatch
          the collectives)."""
        if self.force_overlap:
            return True
        if not self.overlap_allreduce or (dist.get_backend(self.group) != "nccl" and not self.probe_any_backend):
            return False
        st = self.__dict__.setdefault("_ov_probe", {"n": 0, "t": {False: [], True: []}, "use": None, "ev": None})
        if st["use"] is not None:
            return st["use"]
        mode = st["n"] >= 2
        st["mode"] = mode
        st["ev"] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        st["ev"][0].record()
        return mode

    def _probe_end(self):
        st = self.__dict__.get("_ov_probe")
        if not st or st["use"] is not None or st["ev"] is None:
            return
        st["ev"][1].record()
        st["ev"][1].synchronize()
        st["t"][st["mode"]].append(st["ev"][0].elapsed_time(st["ev"][1]))
        st["ev"] = None
        st["n"] += 1
        if st["n"] == 4:
            t = torch.tensor([min(st["t"][False]), min(st["t"][True])], device=self.grads.flat.device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            plain, over = float(t[0]), float(t[1])
            st["use"] = over <= self.overlap_min_gain * plain
            st["measured_ms"] = {"plain": plain, "overlapped": over}
            if not st["use"] and over > 1.25 * plain:      # not merely "no gain" but clearly worse: worth a warning
                import warnings
                warnings.warn(f"FusedStep: the overlapped gradient all-reduce measured {over:.2f} ms per step against {plain:.2f} ms "
                              "for one message after the backward -- using the plain form (BSMS_OVERLAP_ALLREDUCE=0 skips this check)",
                              RuntimeWarning)

    def _bucket_schedule(self, depth):
        """Which `block_done_events` entry of bsms_bsgmp_bwd_ev releases each bucket of `self.grads`.
        The backward produces the weight gradients in this order: decoder (deferred onto side lane 0 in front of the U-Net),
        the U-Net blocks in EXECUTION order e = 0..2L -- up_gmps[L-1] .. up_gmps[0], bottom_gmp, down_gmps[L-1] .. down_gmps[0]
        -- and the encoder (on the caller's stream, after which the lanes are joined).  The flat gradient buffer is laid out
        in reversed parameter order (dp.GradBuckets), i.e. roughly in this order too, so consecutive buckets complete
        one after the other.  A bucket is released by the LAST stage any of its parameters belongs to; `None` = only the
        final join (encoder, and whatever shares a bucket with it)."""
        m, L = self.model, depth
        stage = {}
        for q in m.decode.flat_params():
            stage[q] = 0                                          # covered by the first block's event (in-order lanes)
        blocks = [*m.process.down_gmps, m.process.bottom_gmp, *m.process.up_gmps]
        for k, blk in enumerate(blocks):
            e = (2 * L - k) if k < L else (L if k == L else L - 1 - (k - (L + 1)))     # storage index -> execution index
            for q in (*blk.mlp_node.flat_params(), *blk.mlp_edge.flat_params()):
                stage[q] = e
        out = []
        for bk in self.grads.buckets:
            st = [stage.get(q) for q in bk["params"]]
            out.append(None if any(v is None for v in st) else max(st))
        return out

    def _backward_overlapped(self, b, tar, mask, ews, B, N):
        """The backward with the gradient all-reduce issued PER BUCKET from a communication stream that waits for the
        side-lane event after which the bucket's gradients are final -- the ring transfer of the decoder / up-path buckets
        runs under the down-path blocks; only the last bucket (first down block + encoder) is exposed after the join.
        The host enqueues the whole backward first (it runs ~4 ms ahead of the GPU), then the waits + all-reduces: they
        execute on the GPU as soon as their event fires, not when the host gets there.  Every rank issues the same
        collectives in the same order (the schedule depends on the model only); sums are bitwise rank-independent."""
        depth = b["depth"]
        if self._overlap is None or self._overlap["depth"] != depth:
            evs = [torch.cuda.Event() for _ in range(2 * depth + 1)]
            for e in evs:
                e.record()                     # materialises the hipEvent_t (torch creates it on first use)
            sched = self._bucket_schedule(depth)
            used = {e for e in sched if e is not None}
            ptrs = [evs[i].cuda_event if i in used else None for i in range(2 * depth + 1)]
            self._overlap = dict(depth=depth, events=evs, sched=sched, ptrs=_abi.ptr_array(ptrs))
            self._comm = torch.cuda.Stream(device=b["h0"].device)
        ov = self._overlap
        self._backward(b, tar, mask, ews, B, N, events=ov["ptrs"])        # ends with bsms_side_lanes_join on the caller's stream
        main = torch.cuda.current_stream()
        works = []
        for k, e in self._issue_order(ov["sched"]):                       # identical on every rank: a function of the model only
            bk = self.grads.buckets[k]
            if e is None:                                                    # what only the final join releases: caller's stream
                works.append(dist.all_reduce(bk["view"], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                continue
            with torch.cuda.stream(self._comm):
                self._comm.wait_event(ov["events"][e])
                works.append(dist.all_reduce(bk["view"], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()                           # the caller's stream waits for the collectives (no host block with nccl)
        main.wait_stream(self._comm)

    @staticmethod
    def _issue_order(sched):
        """Order in which the bucket all-reduces are issued: [(bucket index, releasing event or None)].  First the buckets an
        event releases, in bucket order (= the order their events fire: the flat buffer is laid out in backward order, and
        `_bucket_schedule` makes the event index non-decreasing along it), then the buckets only the final join releases.
        A pure function of the schedule, which is a pure function of the model: every rank issues the same collectives in
        the same order (tests/test_dp_gloo.py::test_overlapped_allreduce_issue_order_is_rank_independent)."""
        first = [(k, e) for k, e in enumerate(sched) if e is not None]
        return first + [(k, None) for k, e in enumerate(sched) if e is None]

    def prediction(self):
        """[B,N,C] prediction of the last step (a static buffer: clone it to keep it)."""
        return self._buf["pred"]

    def _replay(self, b, node_in, tar, mask, ews, B, N, world):
        if b["in_static"] is None:
            b["in_static"] = (torch.empty_like(node_in), torch.empty_like(tar), torch.empty_like(mask))
            self._graphs = None
        sn, st, sm = b["in_static"]
        sn.copy_(node_in); st.copy_(tar); sm.copy_(mask)
        if self._graphs is None:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):             # warm-up outside capture: workspaces, lazy kernel attributes, lanes
                self._forward(b, sn, st, sm, ews, B, N)
                self._backward(b, st, sm, ews, B, N)
            torch.cuda.current_stream().wait_stream(side)
            gf, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            # thread_local: a trainer.DevicePrefetcher thread may be allocating / uploading the NEXT batch's plans right
            # now (hipMalloc, hipMemcpyAsync on its own streams); in the default global mode such a call from another
            # thread invalidates this capture
            with torch.cuda.graph(gf, capture_error_mode="thread_local"):
                self._forward(b, sn, st, sm, ews, B, N)
            with torch.cuda.graph(gb, capture_error_mode="thread_local"):
                self._backward(b, st, sm, ews, B, N)
            self._graphs = (gf, gb, ews)              # the graphs hold raw pointers: keep what they point at alive
        gf, gb, _ = self._graphs
        gf.replay()
        if world > 1:
            dist.all_reduce(b["sums"], op=dist.ReduceOp.SUM, group=self.group)
        gb.replay()
        if world > 1:
            dist.all_reduce(self.grads.flat, op=dist.ReduceOp.SUM, group=self.group)
        return b["loss"][0].clone()
