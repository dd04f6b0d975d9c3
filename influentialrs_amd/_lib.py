"""ctypes binding of libirs_hip.so (include/irs_hip.h).

The product path has no CPU fallback: if the shared library is missing or
cannot be loaded, every entry raises.  `load()` does not touch the GPU, so the
symbol table can be checked on a CPU-only box.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libirs_hip.so")

IRS_MASK_IRN, IRS_MASK_CAUSAL = 0, 1
IRS_SWEEP_BF16, IRS_SWEEP_F32, IRS_SWEEP_EXHAUSTIVE = 0, 1, 2
IRS_ROW_FALLBACK, IRS_ROW_NO_CANDIDATE, IRS_ROW_FEWER_THAN_K = 1, 2, 4
IRS_GEMM_F32, IRS_GEMM_X6, IRS_GEMM_H3 = 0, 1, 2
IRS_PROF_NONE, IRS_PROF_LINEAR, IRS_PROF_ATTN, IRS_PROF_SWEEP, IRS_PROF_REFINE, IRS_PROF_SWEEP_EMIT, IRS_PROF_LAYER = 0, 1, 2, 3, 4, 5, 6


class IrsDims(ctypes.Structure):
    _fields_ = [("n_item", c_int64), ("n_user", c_int64), ("d", c_int32), ("max_len", c_int32),
                ("n_heads", c_int32), ("ffn_dim", c_int32), ("n_layers", c_int32), ("u_dim", c_int32),
                ("mask_mode", c_int32), ("max_rows", c_int32), ("max_k", c_int32), ("max_seqs", c_int32)]


class IrsShard(ctypes.Structure):
    _fields_ = [("rank", c_int32), ("world", c_int32), ("item_lo", c_int64), ("item_hi", c_int64)]


# name -> (restype, argtypes); exactly the symbols include/irs_hip.h declares
SIGNATURES = {
    "irs_abi_version": (c_int32, []),
    "irs_last_error": (c_char_p, [c_void_p]),
    "irs_create": (c_int32, [POINTER(c_void_p), POINTER(IrsDims), POINTER(IrsShard)]),
    "irs_destroy": (None, [c_void_p]),
    "irs_bind_weight": (c_int32, [c_void_p, c_char_p, c_void_p, c_int64]),
    "irs_derived_bytes": (c_size_t, [c_void_p]),
    "irs_finalize_weights": (c_int32, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "irs_workspace_bytes": (c_size_t, [c_void_p]),
    "irs_bind_workspace": (c_int32, [c_void_p, c_void_p, c_size_t]),
    "irs_pif": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "irs_decode": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "irs_score_topk": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "irs_score_topk_carry": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "irs_score_gather": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p]),
    "irs_score_count_before": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "irs_score_dense": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_void_p]),
    "irs_score_lse": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "irs_score_topk_lse": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p]),
    "irs_ce_forward": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "irs_ce_grad_logits": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, ctypes.c_float, c_void_p, c_int64, c_void_p]),
    "irs_build_eval_batch": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int64,
                                       c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "irs_merge_topk": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "irs_pack_topk": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "irs_merge_topk_keys": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "irs_path_step": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_void_p,
                                c_int32, c_int32, c_int32, c_uint64, c_void_p, c_void_p]),
    "irs_generate_paths": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                     c_int32, c_uint64, c_int32, c_void_p, c_void_p, c_void_p]),
    "irs_beam_step": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    "irs_beam_search": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                  c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "irs_comm_unique_id": (c_int32, [c_void_p]),
    "irs_comm_init_rccl": (c_int32, [POINTER(c_void_p), c_void_p, c_int32, c_int32]),
    "irs_comm_init_callbacks": (c_int32, [POINTER(c_void_p), c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "irs_comm_destroy": (None, [c_void_p]),
    "irs_comm_last_error": (c_char_p, []),
    "irs_comm_is_rccl": (c_int32, [c_void_p]),
    "irs_comm_exchange_kind": (c_int32, [c_void_p]),
    "irs_comm_rccl_version": (c_int32, []),
    "irs_allgather_rows": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "irs_exchange_topk": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "irs_generate_paths_sharded": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                             c_int32, c_int32, c_uint64, c_int32, c_void_p, c_void_p, c_void_p]),
    "irs_beam_search_sharded": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                          c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "irs_sharded_graph_state": (c_int32, [c_void_p]),
    "irs_set_sharded_overlap": (c_int32, [c_void_p, c_int32]),
    "irs_get_sharded_overlap": (c_int32, [c_void_p]),
    "irs_set_decoder_gemm": (c_int32, [c_void_p, c_int32]),
    "irs_get_decoder_gemm": (c_int32, [c_void_p]),
    "irs_get_decoder_gemm_effective": (c_int32, [c_void_p]),
    "irs_set_decoder_seq": (c_int32, [c_void_p, c_int32]),
    "irs_get_decoder_seq": (c_int32, [c_void_p]),
    "irs_decoder_seq_last": (c_int32, [c_void_p]),
    "irs_debug_ptr": (c_void_p, [c_void_p, c_int32]),
    "irs_h3_range_bound": (c_float, [c_void_p]),
    "irs_prof_enable": (c_int32, [c_void_p, c_int32]),
    "irs_prof_read": (c_int32, [c_void_p, POINTER(c_int32), POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
}

# collective callbacks of irs_comm_init_callbacks (include/irs_hip.h): (user, send, recv, bytes_per_rank, stream) -> int
ALLGATHER_FN = ctypes.CFUNCTYPE(c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p)
ALLTOALL_FN = ctypes.CFUNCTYPE(c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p)
ALLREDUCE_F32_FN = ctypes.CFUNCTYPE(c_int32, c_void_p, c_void_p, c_size_t, c_int32, c_void_p)
IRS_COMM_ID_BYTES = 128

_LIB = None


class IrsLibraryError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """dlopen the in-tree library and set prototypes.  Raises IrsLibraryError if
    it is missing (build with `python -m influentialrs_amd.build`)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise IrsLibraryError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -m influentialrs_amd.build` or __graft_entry__.build().")
    # One HIP runtime per process: torch ships its own libamdhip64 and must load it
    # first so that this library's DT_NEEDED entry resolves to the same copy
    # (two runtimes -> "no ROCm-capable device" in whichever came second).
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise IrsLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.irs_abi_version() != 1:
        raise IrsLibraryError("libirs_hip.so ABI version mismatch")
    _LIB = lib
    return lib
