"""ctypes binding of libgnnrag_hip.so (C ABI declared in include/gnnrag.h).

There is deliberately NO fallback: if the HIP library is missing or an entry point
is absent, importing/using the product path raises.  (CPU restatements live under
``oracle/`` and are test infrastructure only.)
"""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "lib", "libgnnrag_hip.so")

c_i32p = C.POINTER(C.c_int32)
c_f32p = C.POINTER(C.c_float)


class CsrStruct(C.Structure):
    """Mirror of ``struct gnnrag_csr`` (include/gnnrag.h)."""
    _fields_ = [
        ("B", C.c_int32), ("N", C.c_int32), ("R1", C.c_int32), ("heavy_deg", C.c_int32),
        ("F", C.c_int64),
        ("row_ptr", C.c_void_p * 2), ("edge", C.c_void_p * 2), ("perm", C.c_void_p * 2),
        ("w_gnn", C.c_void_p * 2), ("w_rel", C.c_void_p * 2),
        ("heavy", C.c_void_p * 2), ("chunk_off", C.c_void_p * 2),
        ("n_heavy", C.c_void_p), ("n_chunks", C.c_void_p),
        ("heavy_cap", C.c_int32), ("max_chunks", C.c_int32),
        ("big_cnt", C.c_void_p), ("big_nodes", C.c_void_p),
        ("big_deg", C.c_int32), ("hub_sorted", C.c_int32),
        ("edge_l", C.c_void_p * 2), ("rel_off", C.c_void_p), ("rel_rows", C.c_void_p),
        ("rel_total", C.c_int32), ("rel_max", C.c_int32),
        ("edge_m", C.c_void_p), ("m_from", C.c_void_p), ("m_dst", C.c_void_p),
        ("hub_q_off", C.c_void_p * 2), ("hub_wbase", C.c_void_p * 2),
    ]


class RelorderStruct(C.Structure):
    """Mirror of ``struct gnnrag_relorder`` (include/gnnrag.h)."""
    _fields_ = [("F", C.c_int64), ("rel_total", C.c_int32), ("n_chunks", C.c_int32),
                ("ht", C.c_void_p), ("perm", C.c_void_p), ("w", C.c_void_p), ("row_ptr", C.c_void_p),
                ("chunk_ptr", C.c_void_p)]


class LayerParams(C.Structure):
    """Mirror of ``struct gnnrag_layer_params`` (include/gnnrag.h)."""
    _fields_ = [("W_rel", C.c_void_p), ("b_rel", C.c_void_p), ("pos_fwd", C.c_void_p), ("pos_inv", C.c_void_p),
                ("W_e2e", C.c_void_p), ("b_e2e", C.c_void_p)]


# name -> (restype, argtypes); every symbol include/gnnrag.h declares
_VP = C.c_void_p
SIGNATURES = {
    "gnnrag_csr_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int, C.c_int]),
    "gnnrag_csr_scratch_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "gnnrag_csr_build": (C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                   _VP, C.c_size_t, _VP, C.c_size_t, C.POINTER(CsrStruct), _VP]),
    "gnnrag_csr_build_counts": (C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_int32, _VP, C.c_size_t, _VP, C.c_size_t, C.POINTER(CsrStruct), _VP]),
    "gnnrag_csr_status": (C.c_int, [C.POINTER(CsrStruct), _VP]),
    "gnnrag_csr_concat": (C.c_int, [C.POINTER(C.POINTER(CsrStruct)), C.c_int32, C.c_int32, C.c_int32, _VP, C.c_size_t,
                                    C.POINTER(CsrStruct), _VP]),
    "gnnrag_narrow_tuple": (C.c_int, [_VP, _VP, _VP, C.c_int64, _VP, C.c_int32]),
    "gnnrag_csr_permute_weight": (C.c_int, [C.POINTER(CsrStruct), _VP, C.c_int, _VP, _VP, _VP]),
    "gnnrag_linear": (C.c_int, [_VP, C.c_int64, C.c_int32, _VP, _VP, _VP, C.c_int64, C.c_int, _VP,
                                C.c_int32, C.c_int32, _VP]),
    "gnnrag_linear_pair": (C.c_int, [_VP, _VP, C.c_int64, C.c_int32, _VP, _VP, _VP, _VP, C.c_int64, _VP, _VP,
                                     C.c_int32, C.c_int32, _VP]),
    "gnnrag_aggregate_workspace_bytes": (C.c_size_t, [C.POINTER(CsrStruct), C.c_int32, C.c_int32]),
    "gnnrag_aggregate": (C.c_int, [C.POINTER(CsrStruct), _VP, _VP, _VP, _VP, _VP, C.c_int32, C.c_int32,
                                   _VP, C.c_size_t, _VP]),
    "gnnrag_aggregate_fused": (C.c_int, [C.POINTER(CsrStruct), _VP, _VP, _VP, C.c_int32, _VP, C.c_size_t, _VP]),
    "gnnrag_aggregate_fused_variant": (C.c_int, [C.POINTER(CsrStruct), C.c_int32]),
    "gnnrag_aggregate_fused_hub_form": (C.c_int, [C.POINTER(CsrStruct), C.c_int32, _VP, C.c_size_t, _VP, _VP]),
    "gnnrag_relorder_bytes": (C.c_size_t, [C.POINTER(CsrStruct), C.c_int]),
    "gnnrag_relorder_scratch_bytes": (C.c_size_t, [C.POINTER(CsrStruct)]),
    "gnnrag_relorder_build": (C.c_int, [C.POINTER(CsrStruct), _VP, _VP, _VP, _VP, _VP, C.c_size_t, _VP, C.c_size_t,
                                        C.POINTER(RelorderStruct), _VP]),
    "gnnrag_backward_workspace_bytes": (C.c_size_t, [C.POINTER(CsrStruct), C.POINTER(RelorderStruct), C.c_int32,
                                                     C.c_int32]),
    "gnnrag_aggregate_fused_backward": (C.c_int, [C.POINTER(CsrStruct), C.POINTER(RelorderStruct), _VP, _VP, _VP, _VP, _VP,
                                                 C.c_int32, _VP, C.c_size_t, _VP]),
    "gnnrag_aggregate_backward": (C.c_int, [C.POINTER(CsrStruct), C.POINTER(RelorderStruct), _VP, _VP, _VP, _VP, _VP,
                                            _VP, _VP, _VP, _VP, C.c_int32, C.c_int32, _VP, C.c_size_t, _VP]),
    "gnnrag_typelayer_backward": (C.c_int, [C.POINTER(CsrStruct), C.POINTER(RelorderStruct), _VP, _VP, C.c_int, _VP,
                                            C.c_int32, _VP, C.c_size_t, _VP]),
    "gnnrag_lstm_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "gnnrag_lstm_forward": (C.c_int, [_VP] * 10 + [C.c_int32] * 4 + [_VP, C.c_size_t, _VP]),
    "gnnrag_seed_retrieve": (C.c_int, [_VP, _VP, _VP, C.c_int32, C.c_int32, C.c_int32, _VP]),
    "gnnrag_query_reform": (C.c_int, [_VP, _VP, _VP, C.c_int64, _VP, _VP, _VP, C.c_int32, C.c_int32, C.c_int32, _VP]),
    "gnnrag_topp_candidates": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32, C.c_double, C.c_double, _VP, _VP, _VP]),
    "gnnrag_topp_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "gnnrag_topp_candidates_ws": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32, C.c_double, C.c_double, _VP, _VP, _VP,
                                            C.c_size_t, _VP]),
    "gnnrag_relation_tables": (C.c_int, [C.POINTER(CsrStruct), _VP, _VP, _VP, _VP, _VP, C.c_int32, C.c_int32,
                                         C.c_int32, _VP]),
    "gnnrag_update_score_fused": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, C.c_int32,
                                            C.c_int32, C.c_int32, _VP]),
    "gnnrag_update_score": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, C.c_int32,
                                      C.c_int32, C.c_int32, _VP]),
    "gnnrag_masked_softmax": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32, _VP]),
    "gnnrag_typelayer": (C.c_int, [C.POINTER(CsrStruct), _VP, C.c_int, _VP, C.c_int32, _VP, C.c_size_t, _VP]),
    "gnnrag_layer_workspace_bytes": (C.c_size_t, [C.POINTER(CsrStruct), C.c_int32, C.c_int32]),
    "gnnrag_stack_workspace_bytes": (C.c_size_t, [C.POINTER(CsrStruct), C.c_int32, C.c_int32, C.c_int32]),
    "gnnrag_rel_transform": (C.c_int, [_VP, _VP, C.c_int64, C.c_int32, C.c_int32, C.POINTER(LayerParams), C.c_int32, _VP,
                                      _VP, _VP]),
    "gnnrag_gemm_tn_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "gnnrag_gemm_tn": (C.c_int, [_VP, _VP, C.c_int64, C.c_int32, C.c_int32, _VP, _VP, C.c_size_t, _VP]),
    "gnnrag_rel_planes_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "gnnrag_relation_tables_planes": (C.c_int, [C.POINTER(CsrStruct), _VP, _VP, _VP, _VP, C.c_int32, C.c_int32, _VP]),
    "gnnrag_reason_layer": (C.c_int, [C.POINTER(CsrStruct)] + [_VP] * 9 + [C.c_int32] + [_VP] * 8 +
                            [_VP, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _VP]),
    "gnnrag_reason_stack": (C.c_int, [C.POINTER(CsrStruct), C.c_int32, C.POINTER(LayerParams)] + [_VP] * 5 +
                            [C.c_int32] + [_VP] * 6 + [_VP, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _VP]),
    "gnnrag_reason_stack_capture": (C.c_int, [C.POINTER(CsrStruct), C.c_int32, C.POINTER(LayerParams)] + [_VP] * 5 +
                                    [C.c_int32] + [_VP] * 6 + [_VP, C.c_size_t, C.c_int32, C.c_int32, C.c_int32,
                                                               C.c_int32, _VP, C.POINTER(C.c_void_p)]),
    "gnnrag_graph_launch": (C.c_int, [_VP, _VP]),
    "gnnrag_graph_destroy": (C.c_int, [_VP]),
    "gnnrag_frontier_workspace_bytes": (C.c_size_t, [C.POINTER(CsrStruct)]),
    "gnnrag_frontier_supported": (C.c_int, [C.POINTER(CsrStruct), C.c_int32]),
    "gnnrag_frontier_build": (C.c_int, [C.POINTER(CsrStruct), _VP, _VP, C.c_size_t, _VP]),
    "gnnrag_relation_tables_frontier": (C.c_int, [C.POINTER(CsrStruct), _VP, _VP, _VP, _VP, _VP, _VP, C.c_int32,
                                                  C.c_int32, _VP]),
    "gnnrag_aggregate_fused_frontier": (C.c_int, [C.POINTER(CsrStruct), _VP, _VP, _VP, _VP, C.c_int32, _VP]),
    "gnnrag_frontier_read": (C.c_int, [C.POINTER(CsrStruct), _VP, _VP, _VP, _VP]),
    "gnnrag_stream_copy": (C.c_int, [_VP, _VP, C.c_int64, _VP]),
    "gnnrag_abi_version": (C.c_int, []),
    "gnnrag_error_string": (C.c_char_p, [C.c_int]),
}

ABI_VERSION = 16
PATH_AUTO, PATH_UNFUSED, PATH_FUSED = 0, 1, 2
PATH_ONLY_FWD, PATH_ONLY_INV = 0x10, 0x20        # OR-ed into the path: one-direction layers (NSM)
PATH_SEED_PRIOR = 0x40                           # OR-ed into the path: the (first layer's) prior is a seed distribution
PATH_REUSE_PROJ = 0x80                           # gnnrag_reason_stack: the workspace still holds the relation projections
E_TUPLE = -4
_lib = None


class GnnragError(RuntimeError):
    pass


def load():
    """Loads the library once; raises if it is not built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("GNNRAG_LIB", LIB_PATH)     # experiment builds (tools/tune_variants.py)
    if not os.path.exists(path):
        raise GnnragError(
            "HIP extension %s is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.gnnrag_abi_version() != ABI_VERSION:
        raise GnnragError("ABI mismatch: library %d, binding %d" % (lib.gnnrag_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(code: int, what: str = ""):
    if code != 0:
        msg = load().gnnrag_error_string(code)
        raise GnnragError("%s failed (%d): %s" % (what or "gnnrag call", code,
                                                  msg.decode() if msg else "?"))
