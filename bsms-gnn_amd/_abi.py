"""ctypes binding of libbsms_hip.so (include/bsms_hip.h).  No torch types cross this boundary: only
raw device pointers, sizes and the HIP stream handle.  There is NO CPU fallback: if the library is
missing or a call fails, this raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbsms_hip.so")

c_i64, c_int, c_void_p, c_size_t = C.c_int64, C.c_int, C.c_void_p, C.c_size_t
PP = C.POINTER(c_void_p)

# name -> (restype, argtypes); one entry per declaration in include/bsms_hip.h
SIGNATURES = {
    "bsms_abi_version": (c_int, []),
    "bsms_last_error": (C.c_char_p, []),
    "bsms_plan_create": (c_int, [c_void_p, c_i64, c_i64, PP]),
    "bsms_plan_set_pool": (c_int, [c_void_p, c_void_p, c_i64]),
    "bsms_plan_bind_edge_weights": (c_int, [c_void_p, c_void_p, c_void_p]),
    "bsms_plan_bound_edge_weights": (c_void_p, [c_void_p]),
    "bsms_plan_destroy": (c_int, [c_void_p]),
    "bsms_plan_pool_trim": (c_int, []),
    "bsms_plan_num_nodes": (c_i64, [c_void_p]),
    "bsms_plan_num_edges": (c_i64, [c_void_p]),
    "bsms_plan_num_pooled": (c_i64, [c_void_p]),
    "bsms_plan_min_out_degree": (c_i64, [c_void_p]),
    "bsms_plan_max_source": (c_i64, [c_void_p]),
    "bsms_plan_export": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bsms_plan_export_ex": (c_i64, [c_void_p, c_int, c_void_p]),
    "bsms_plan_concat": (c_int, [PP, c_int, c_void_p, c_void_p, c_void_p, c_void_p, PP]),
    "bsms_segment_sum_fwd": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_int, c_void_p, c_void_p]),
    "bsms_segment_sum_bwd": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_void_p]),
    "bsms_segment_sum_bf16": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_void_p]),
    "bsms_cal_ew": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bsms_edge_conv": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "bsms_scatter_rows": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_void_p, c_i64, c_void_p, c_void_p]),
    "bsms_gather_rows": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_void_p, c_i64, c_void_p, c_void_p]),
    "bsms_mlp_saved_bytes": (c_size_t, [c_i64, c_i64, c_i64, c_i64, c_int]),
    "bsms_mlp_work_bytes": (c_size_t, [c_i64, c_i64, c_i64, c_i64, c_int]),
    "bsms_mlp_fwd": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_int, PP, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bsms_mlp_fwd_ex": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_int, PP, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "bsms_mlp_bwd": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_int, PP, c_void_p, c_void_p,
                             c_void_p, PP, c_void_p]),
    "bsms_mlp_bwd_ex": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_int, PP, c_void_p, c_void_p,
                                c_void_p, PP, c_int, c_void_p]),
    "bsms_gmp_saved_bytes": (c_size_t, [c_i64, c_i64, c_i64, c_i64, c_int]),
    "bsms_gmp_work_bytes": (c_size_t, [c_i64, c_i64, c_i64, c_i64, c_int]),
    "bsms_gmp_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, PP, c_void_p, c_void_p,
                             c_void_p, c_void_p]),
    "bsms_gmp_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, PP, c_void_p,
                             c_void_p, c_void_p, PP, c_void_p]),
    "bsms_bsgmp_saved_bytes": (c_size_t, [PP, c_int, c_i64, c_i64, c_i64, c_int]),
    "bsms_bsgmp_work_bytes": (c_size_t, [PP, c_int, c_i64, c_i64, c_i64, c_int]),
    "bsms_bsgmp_infer_work_bytes": (c_size_t, [PP, c_int, c_i64, c_i64, c_i64, c_int]),
    "bsms_bsgmp_fwd": (c_int, [PP, PP, c_int, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, PP, c_void_p, c_void_p,
                               c_void_p, c_void_p]),
    "bsms_bsgmp_fwd_ex": (c_int, [PP, PP, c_int, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, PP, c_void_p, c_void_p,
                                  c_void_p, c_int, c_void_p]),
    "bsms_bsgmp_saved_bytes_p": (c_size_t, [PP, c_int, c_i64, c_i64, c_i64, c_int, c_int]),
    "bsms_bsgmp_fwd_p": (c_int, [PP, PP, c_int, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, PP, c_void_p, c_void_p,
                                 c_void_p, c_int, c_int, c_void_p]),
    "bsms_bsgmp_bwd_p": (c_int, [PP, PP, c_int, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, PP, c_void_p,
                                 c_void_p, c_void_p, PP, c_int, c_void_p]),
    "bsms_bsgmp_bwd": (c_int, [PP, PP, c_int, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, PP, c_void_p,
                               c_void_p, c_void_p, PP, c_void_p]),
    "bsms_bsgmp_bwd_ex": (c_int, [PP, PP, c_int, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, PP, c_void_p,
                                  c_void_p, c_void_p, PP, c_int, c_int, c_void_p]),
    "bsms_bsgmp_bwd_ev": (c_int, [PP, PP, c_int, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, PP, c_void_p,
                                  c_void_p, c_void_p, PP, c_int, c_int, PP, c_void_p]),
    "bsms_side_lanes_join": (c_int, [c_void_p]),
    "bsms_streams_overlap": (c_int, [c_void_p, c_void_p]),
    "bsms_sim_work_bytes": (c_size_t, [c_i64]),
    "bsms_sim_prologue": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bsms_sim_epilogue": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bsms_sim_loss_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
    "bsms_hierarchy_create": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_i64, c_int, PP]),
    "bsms_hierarchy_create_f32": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_i64, c_int, PP]),
    "bsms_hierarchy_destroy": (c_int, [c_void_p]),
    "bsms_hierarchy_level_nodes": (c_i64, [c_void_p, c_int]),
    "bsms_hierarchy_level_edges": (c_i64, [c_void_p, c_int]),
    "bsms_hierarchy_copy_edges": (c_int, [c_void_p, c_int, c_void_p]),
    "bsms_hierarchy_copy_ids": (c_int, [c_void_p, c_int, c_void_p]),
    "bsms_adamw_work_bytes": (c_size_t, []),
    "bsms_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, C.c_float, C.c_float, C.c_float, C.c_float,
                                C.c_float, c_i64, C.c_float, c_void_p, c_void_p, c_void_p]),
}

_ERRORS = {-1: "BSMS_E_INVALID_ARG", -2: "BSMS_E_SHAPE", -3: "BSMS_E_UNSUPPORTED", -4: "BSMS_E_HIP"}
_lib = None


class BsmsError(RuntimeError):
    pass


def lib():
    """Load the shared library (once).  Raises if it has not been built -- the product path never
    silently degrades to a PyTorch implementation."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BsmsError(f"{LIB_PATH} not found: build it with `python bsms-gnn_amd/build.py` "
                            "(or __graft_entry__.build()); there is no CPU/PyTorch fallback")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().bsms_last_error().decode(errors="replace")
        raise BsmsError(f"{what}: {_ERRORS.get(rc, rc)}: {msg}")


def ptr_array(ptrs):
    arr = (c_void_p * len(ptrs))(*ptrs)
    return C.cast(arr, PP), arr  # keep `arr` alive while the call runs
