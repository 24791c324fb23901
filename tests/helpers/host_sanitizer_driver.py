#!/usr/bin/env python3
"""Driver for the ASan / UBSan build of the library's HOST code (csrc/plan.hip, csrc/hierarchy.hip:
bsms-gnn_amd/_build/asan/libbsms_host_asan.so, built by build.py: build_host_sanitized).  Runs under
LD_PRELOAD=<libclang_rt.asan> in a process WITHOUT torch (ctypes + numpy only, so the sanitizer sees our code and
the HIP runtime, not a 1 GB framework): the hierarchy builder on meshes with several components / quads / float32
positions, and -- when a GPU is present -- the plan life cycle the round-3 training loop exercises: create, set_pool,
re-pool, export, destroy into the recycling pool, re-create from recycled blocks of other sizes, concurrent creation from
threads, pool trim.  Any sanitizer report makes the process exit non-zero (halt_on_error) with the report on stderr.

usage: host_sanitizer_driver.py <path to the sanitized .so> [--gpu]"""
import ctypes as C
import sys
import threading

import numpy as np


def tri_mesh(n, seed):
    rng = np.random.default_rng(seed)
    nx = int(np.sqrt(n))
    xs, ys = np.meshgrid(np.arange(nx), np.arange(nx), indexing="ij")
    idx = xs * nx + ys
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    cells = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)])
    pairs = np.concatenate([cells[:, [0, 1]], cells[:, [1, 2]], cells[:, [2, 0]]])
    und = np.unique(np.stack([pairs.max(1), pairs.min(1)], 1), axis=0)
    coo = np.ascontiguousarray(np.stack([np.concatenate([und[:, 0], und[:, 1]]), np.concatenate([und[:, 1], und[:, 0]])]).astype(np.int64))
    pos = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.float64) + 0.3 * rng.random((nx * nx, 2))
    return coo, pos, nx * nx


def main():
    lib = C.CDLL(sys.argv[1])
    gpu = "--gpu" in sys.argv
    lib.bsms_last_error.restype = C.c_char_p
    vp, i64 = C.c_void_p, C.c_int64
    lib.bsms_hierarchy_create.argtypes = [vp, i64, i64, vp, i64, C.c_int, C.POINTER(vp)]
    lib.bsms_hierarchy_create_f32.argtypes = [vp, i64, i64, vp, i64, C.c_int, C.POINTER(vp)]
    lib.bsms_hierarchy_destroy.argtypes = [vp]
    for f in (lib.bsms_hierarchy_level_nodes, lib.bsms_hierarchy_level_edges):
        f.argtypes, f.restype = [vp, C.c_int], i64
    lib.bsms_hierarchy_copy_edges.argtypes = [vp, C.c_int, vp]
    lib.bsms_hierarchy_copy_ids.argtypes = [vp, C.c_int, vp]
    lib.bsms_plan_create.argtypes = [vp, i64, i64, C.POINTER(vp)]
    lib.bsms_plan_set_pool.argtypes = [vp, vp, i64]
    lib.bsms_plan_destroy.argtypes = [vp]
    lib.bsms_plan_export.argtypes = [vp, vp, vp, vp, vp]
    for f in (lib.bsms_plan_num_nodes, lib.bsms_plan_num_edges, lib.bsms_plan_num_pooled, lib.bsms_plan_max_source, lib.bsms_plan_min_out_degree):
        f.argtypes, f.restype = [vp], i64

    # ---------------------------------------------------------------- hierarchy builder (pure host)
    levels = []
    for n, seed, depth, f32 in ((400, 0, 3, False), (2500, 1, 5, True), (49, 2, 2, False), (900, 3, 7, True)):
        coo, pos, nn = tri_mesh(n, seed)
        if seed == 2:                                   # two components + an isolated pair: several clusters / seeds
            coo = np.ascontiguousarray(np.concatenate([coo, coo + nn, np.array([[2 * nn, 2 * nn + 1], [2 * nn + 1, 2 * nn]])], 1))
            pos = np.concatenate([pos, pos + 100.0, np.array([[500.0, 500.0], [501.0, 500.0]])])
            nn = 2 * nn + 2
        h = vp()
        p = np.ascontiguousarray(pos.astype(np.float32 if f32 else np.float64))
        rc = (lib.bsms_hierarchy_create_f32 if f32 else lib.bsms_hierarchy_create)(coo.ctypes.data, coo.shape[1], nn, p.ctypes.data, 2, depth, C.byref(h))
        assert rc == 0, lib.bsms_last_error()
        prev = nn
        for l in range(depth + 1):
            nl, el = lib.bsms_hierarchy_level_nodes(h, l), lib.bsms_hierarchy_level_edges(h, l)
            assert nl == prev and el >= 0
            e = np.empty((2, el), np.int64)
            assert lib.bsms_hierarchy_copy_edges(h, l, e.ctypes.data) == 0
            assert el == 0 or (e.min() >= 0 and e.max() < nl)
            ids = None
            if l < depth:
                k = lib.bsms_hierarchy_level_nodes(h, l + 1)
                ids = np.empty(k, np.int64)
                assert lib.bsms_hierarchy_copy_ids(h, l, ids.ctypes.data) == 0
                assert k == 0 or (np.all(np.diff(ids) > 0) and ids[-1] < nl)
                prev = k
            levels.append((np.ascontiguousarray(e), nl, ids))
        assert lib.bsms_hierarchy_level_nodes(h, depth + 1) < 0 and lib.bsms_hierarchy_copy_ids(h, depth, None) != 0   # range / null checks
        lib.bsms_hierarchy_destroy(h)
    # error paths
    h = vp()
    bad = np.array([[0, 5], [1, 0]], np.int64)
    assert lib.bsms_hierarchy_create(bad.ctypes.data, 2, 3, np.zeros((3, 2)).ctypes.data, 2, 2, C.byref(h)) != 0      # index out of range
    assert lib.bsms_hierarchy_create(None, 0, 0, None, 2, 1, C.byref(h)) != 0 or lib.bsms_hierarchy_destroy(h) == 0
    print(f"hierarchy: {len(levels)} levels built, no sanitizer report")

    # ---------------------------------------------------------------- plan life cycle (needs the GPU for its blocks)
    if not gpu:
        p = vp()
        e, n, _ = levels[0]
        rc = lib.bsms_plan_create(e.ctypes.data, e.shape[1], n, C.byref(p))
        print(f"plan_create without a GPU -> rc {rc}: {lib.bsms_last_error().decode()[:80]!r} (error path only)")
        if rc == 0:
            lib.bsms_plan_destroy(p)
        return

    def make(e, n, ids):
        p = vp()
        assert lib.bsms_plan_create(e.ctypes.data, e.shape[1], n, C.byref(p)) == 0, lib.bsms_last_error()
        if ids is not None and len(ids):
            assert lib.bsms_plan_set_pool(p, ids.ctypes.data, len(ids)) == 0, lib.bsms_last_error()
        return p

    def check(p, e, n, ids):
        assert lib.bsms_plan_num_nodes(p) == n and lib.bsms_plan_num_edges(p) == e.shape[1]
        assert lib.bsms_plan_num_pooled(p) == (0 if ids is None else len(ids))
        rp, src, perm, trp = np.empty(n + 1, np.int32), np.empty(e.shape[1], np.int32), np.empty(e.shape[1], np.int32), np.empty(n + 1, np.int32)
        assert lib.bsms_plan_export(p, rp.ctypes.data, src.ctypes.data, perm.ctypes.data, trp.ctypes.data) == 0
        order = np.argsort(e[1], kind="stable")
        assert np.array_equal(perm, order.astype(np.int32)) and np.array_equal(src, e[0][order].astype(np.int32))

    usable = [(e, n, ids) for e, n, ids in levels if e.shape[1] > 0]
    for rnd in range(6):                                 # retire everything each round: the next round recycles blocks of OTHER sizes
        plans = [make(e, n, ids) for e, n, ids in (usable if rnd % 2 == 0 else usable[::-1])]
        for p, (e, n, ids) in zip(plans, usable if rnd % 2 == 0 else usable[::-1]):
            check(p, e, n, ids)
        e, n, ids = usable[0]                            # re-pool an existing plan (round-3 ADVICE: the old block goes to the pool)
        if ids is not None and len(ids) > 2:
            assert lib.bsms_plan_set_pool(plans[0] if rnd % 2 == 0 else plans[-1], ids[: len(ids) // 2].ctypes.data, len(ids) // 2) == 0
        for p in plans:
            assert lib.bsms_plan_destroy(p) == 0
    errs = []

    def worker(k):
        try:
            for r in range(8):
                e, n, ids = usable[(k + r) % len(usable)]
                p = make(e, n, ids)
                check(p, e, n, ids)
                lib.bsms_plan_destroy(p)
        except BaseException as ex:      # noqa: BLE001
            errs.append(ex)
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    assert lib.bsms_plan_pool_trim() == 0
    assert lib.bsms_plan_destroy(None) == 0 and lib.bsms_plan_num_nodes(None) == -1
    print("plans: create / pool / re-pool / export / destroy / recycle / 6 threads / trim, no sanitizer report")


if __name__ == "__main__":
    main()
