"""ctypes binding of libeagle_hip.so (the C ABI declared in include/eagle_hip.h).

The library is mandatory: there is no CPU or eager fallback anywhere in eagle_amd.  If it is
missing (not built) every kernel call raises ``HipLibraryMissing``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_void_p
from typing import Optional

import torch

LIB_PATH = os.environ.get("EAGLE_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libeagle_hip.so")

_ERR = {1: "EAGLE_ERR_SHAPE (operand shapes outside what the kernel assumes)",
        2: "EAGLE_ERR_LAUNCH (HIP launch failed)", 3: "EAGLE_ERR_ARG (null pointer / bad enum)"}


class HipLibraryMissing(RuntimeError):
    pass


class HipKernelError(RuntimeError):
    pass


# name -> (restype, argtypes).  Mirrors include/eagle_hip.h one to one.
P, I, L, F = c_void_p, c_int, c_int64, c_float
SIGNATURES = {
    "eagle_gemm_skinny_bf16": (I, [P, P, L, I, P, I, L, F, P, P, L, P, P, P, P, L, P, L, I, I, I, I, P]),
    "eagle_packed_weight_bytes": (L, [I, I]),
    "eagle_pack_weight_bf16": (I, [P, L, I, I, P, P]),
    "eagle_gemm_rows_workspace_bytes": (L, [I, I, I, I]),
    "eagle_gemm_rows_bf16": (I, [P, P, L, I, P, I, L, P, P, L, P, P, L, P, P, L, P, L, I, I, I, I, P, L, P]),
    "eagle_gemm_nt_bf16": (I, [P, L, P, L, P, L, P, P, L, I, I, I, I, I, P]),
    "eagle_gemm_nt_splitk_workspace_bytes": (L, [I, I, I]),
    "eagle_gemm_nt_splitk_bf16": (I, [P, L, P, L, P, L, P, P, L, I, I, I, I, I, I, P, P]),
    "eagle_gemm_tn_bf16": (I, [P, L, P, L, P, L, I, I, I, I, I, P]),
    "eagle_gemm_nn_bf16": (I, [P, L, P, L, P, L, P, L, I, I, I, I, I, P]),
    "eagle_transpose_bf16": (I, [P, L, P, L, I, I, I, P]),
    "eagle_rmsnorm_fwd_bf16": (I, [P, P, P, P, I, I, F, P]),
    "eagle_rmsnorm_bwd_workspace_bytes": (L, [I, I]),
    "eagle_rmsnorm_bwd_bf16": (I, [P, P, P, P, P, P, P, I, P, I, I, P]),
    "eagle_attn_decode_workspace_bytes": (L, [I, I, I]),
    "eagle_tree_attention_bf16": (I, [P, L, P, I, I, P, P, I, P, P, L, I, P, I, P, I, L, P, L, P, L, P, P,
                                      I, I, I, I, I, I, I, I, I, P]),
    "eagle_attn_train_fwd_bf16": (I, [P, L, P, L, P, I, I, I, I, P]),
    "eagle_attn_train_bwd_bf16": (I, [P, L, P, L, P, L, P, P, P, L, I, I, I, I, P]),
    "eagle_rope_qk_bf16": (I, [P, L, P, P, P, I, L, I, I, I, I, P]),
    "eagle_topk_workspace_bytes": (L, [I, I, I]),
    "eagle_topk_vocab": (I, [P, L, I, I, I, I, P, P, P, P, L, P]),
    "eagle_topk_vocab_grouped": (I, [P, L, I, I, I, I, P, P, P, I, L, P]),
    "eagle_topk_vocab_noalloc": (I, [P, L, I, I, I, I, P, P, P, L, P]),
    "eagle_topk_merge_candidates": (I, [P, I, I, I, P, P, I, L, P]),
    "eagle_lm_head_topk_rows_workspace_bytes": (L, [I, I, I, I]),
    "eagle_lm_head_topk_rows_bf16": (I, [P, P, L, P, I, L, P, L, P, I, I, I, I, P, P, I, L, P, L, P]),
    "eagle_topk_lse_vocab": (I, [P, L, I, I, I, P, P, P, P, P]),
    "eagle_tree_layout": (I, [P, I, I, I, I, P]),
    "eagle_tree_build": (I, [P, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P]),
    "eagle_tree_gather_ids": (I, [P, L, P, I, P, I, P]),
    "eagle_greedy_accept": (I, [P, P, L, P, I, P, I, I, P, P, P, P, P, P, P, P, P, P, I, P, P, P, I, P]),
    "eagle_kv_compact_bf16": (I, [P, L, L, I, I, I, P, I, P, P, I, P, P]),
    "eagle_gather_accepted_bf16": (I, [P, I, I, P, I, P, P, P, P, P, I, P]),
    "eagle_row_lse_bf16": (I, [P, L, I, I, P, P]),
    "eagle_dyn_step": (I, [I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, P]),
    "eagle_dyn_finalize": (I, [I, I, I, P, P, P, P, P, P, P, P, P, I, P]),
    "eagle_dyn_accept": (I, [I, I, P, P, P, P, P, P, P, P, P, P, P, P, I, P]),
    "eagle_ce_stat_bytes": (I, []),
    "eagle_ce_partial_bf16": (I, [P, P, L, I, I, I, P, P, I, P, P]),
    "eagle_ce_combine": (I, [P, I, I, P, P, P, P, P, P]),
    "eagle_fused_ce_smoothl1_fwd_bf16": (I, [P, P, L, I, I, P, P, I, P, P, P, P, P, P, P]),
    "eagle_ce_smoothl1_bwd_bf16": (I, [P, P, L, I, I, P, P, P, F, F, P, L, P, P, I, P, P]),
    "eagle_embed_concat_bf16": (I, [P, P, P, L, P, I, I, I, P]),
    "eagle_silu_mul_fwd_bf16": (I, [P, L, P, L, I, I, P]),
    "eagle_silu_mul_bwd_bf16": (I, [P, L, P, L, P, L, I, I, P]),
    "eagle_colsum_workspace_bytes": (L, [I, I]),
    "eagle_colsum_bf16": (I, [P, L, I, I, P, P, I, P, P]),
    "eagle_adamw_bf16": (I, [P, P, P, P, L, F, F, F, F, F, I, P, F, P]),
    "eagle_grad_norm_workspace_bytes": (L, []),
    "eagle_grad_clip_coef_bf16": (I, [P, L, F, P, P, P]),
    "eagle_clip_value_bf16": (I, [P, L, F, P]),
    "eagle_scale_bf16": (I, [P, L, P, P]),
    "eagle_add_f32_bf16": (I, [P, P, P, L, P]),
    "eagle_make_eagle_input_bf16": (I, [P, P, P, P, P, P, I, I, I, P]),
    "eagle_input_embed_concat_bf16": (I, [P, L, L, P, L, P, I, F, F, ctypes.c_uint64, ctypes.c_uint64, P, P, P, P, I, I, I, P]),
    "eagle_debug_set_gemm8": (None, [I]),
    "eagle_debug_set_skinny_variant": (None, [I, I, I]),
    "eagle_debug_set_rows": (None, [I, I, I]),
    "eagle_debug_set_attn": (None, [I]),
    "eagle_debug_set_topk": (None, [I]),
    "eagle_debug_set_loss": (None, [I]),
}

_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -m eagle_amd.build` "
                "(eagle_amd has no CPU fallback)")
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise HipKernelError("eagle_amd kernels need device tensors (no CPU path)")
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def call(name: str, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise HipKernelError(f"{name} failed: {_ERR.get(rc, rc)}")


def query(name: str, *args):
    return getattr(lib(), name)(*args)
