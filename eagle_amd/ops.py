"""Thin tensor-level wrappers over the C ABI (include/eagle_hip.h).

Every function launches hand-written HIP kernels on the current torch stream; torch is used
only for device memory and streams.  There is no CPU path: non-device tensors raise.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from . import _hip
from ._hip import call, ptr, query, stream

BF16 = torch.bfloat16
EPI_LINEAR, EPI_SILU_MUL = 0, 1
KEEP_RESIDENT = 0x100   # EAGLE_GEMM_KEEP_RESIDENT
W_PACKED = 0x200        # EAGLE_GEMM_W_PACKED


class PackedWeight:
    """eagle_pack_weight_bf16 image of a weight matrix [N, K] (a derived copy laid out for whole-cache-line streaming; the
    row-major parameter stays the state_dict tensor).  ``data`` is the opaque image, ``shape`` the logical (N, K)."""
    __slots__ = ("data", "shape")

    def __init__(self, data: torch.Tensor, N: int, K: int):
        self.data, self.shape = data, (N, K)

    def data_ptr(self) -> int:
        return self.data.data_ptr()

    def numel(self) -> int:
        return self.data.numel()

    def element_size(self) -> int:
        return self.data.element_size()


def pack_weight(W: torch.Tensor) -> PackedWeight:
    """W [N, K] bf16 (rows contiguous, any row stride that is a multiple of 8) -> its packed image."""
    _chk(W, "W")
    N, K = W.shape
    if W.stride(1) != 1 or K % 32:
        raise ValueError("pack_weight: W [N, K] with contiguous rows and K % 32 == 0")
    out = torch.empty(query("eagle_packed_weight_bytes", N, K) // 2, dtype=BF16, device=W.device)
    call("eagle_pack_weight_bf16", ptr(W), W.stride(0), N, K, ptr(out), stream())
    return PackedWeight(out, N, K)


def _chk(t: torch.Tensor, name: str, dtype=BF16):
    if not t.is_cuda:
        raise _hip.HipKernelError(f"{name}: device tensor required (eagle_amd has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")


def require_device(t: torch.Tensor, what: str) -> None:
    """There is no CPU path: every entry into the kernels goes through this check."""
    if not t.is_cuda:
        raise _hip.HipKernelError(f"{what} needs device tensors: eagle_amd has no CPU path")


def linear_skinny(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None,
                  res: Optional[torch.Tensor] = None, W2: Optional[torch.Tensor] = None,
                  idx0: Optional[torch.Tensor] = None, x1: Optional[torch.Tensor] = None,
                  idx1: Optional[torch.Tensor] = None, M: Optional[int] = None,
                  out: Optional[torch.Tensor] = None, ind0: Optional[torch.Tensor] = None,
                  idx0_bstride: int = 0, norm_eps: float = 0.0, resident: bool = False) -> torch.Tensor:
    """out[M,N] = epi(cat(x[idx0], x1[idx1]) @ W^T).  x [*, K0] row-major; W [N, K] row-major or its PackedWeight image
    (same result bit for bit, faster weight stream).
    ind0: double indirection for x rows (see include/eagle_hip.h); norm_eps > 0 fuses the RMSNorm
    row scale (W must then carry the norm weight folded into its columns).  resident: keep W in the Infinity Cache
    (default cache policy) instead of streaming it with non-temporal loads."""
    _chk(x, "x")
    packed = isinstance(W, PackedWeight)
    if packed:
        if W2 is not None and (not isinstance(W2, PackedWeight) or W2.shape != W.shape):
            raise ValueError("W2 must be packed like W")
    else:
        _chk(W, "W")
        if isinstance(W2, PackedWeight):
            raise ValueError("W2 must be packed like W")
    N, K = W.shape
    K0 = x.shape[-1]
    if x1 is not None:
        _chk(x1, "x1")
        if K0 + x1.shape[-1] != K:
            raise ValueError(f"cat width {K0}+{x1.shape[-1]} != K={K}")
    elif K0 != K:
        raise ValueError(f"x width {K0} != K={K}")
    if M is None:
        M = idx0.numel() if idx0 is not None else x.shape[0]
    if x.stride(-1) != 1 or (not packed and (W.stride(1) != 1 or W.stride(0) != K)):
        raise ValueError("x rows and W must be contiguous along K")
    for i in (idx0, idx1, ind0):
        if i is not None and (i.dtype != torch.int64 or (not i.is_contiguous() and not (i is idx0 and ind0 is not None))):
            raise TypeError("row index arrays must be contiguous int64")
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=x.device)
    epi = (EPI_SILU_MUL if W2 is not None else EPI_LINEAR) | (KEEP_RESIDENT if resident else 0) | (W_PACKED if packed else 0)
    call("eagle_gemm_skinny_bf16", ptr(x), ptr(idx0), x.stride(0) if x.dim() > 1 else K0, K0,
         ptr(ind0), (ind0.numel() if ind0 is not None else 0), idx0_bstride, float(norm_eps),
         ptr(x1), ptr(idx1), (x1.stride(0) if x1 is not None else 0),
         ptr(W.data if packed else W), ptr(W2.data if (packed and W2 is not None) else W2), ptr(bias), ptr(res),
         (res.stride(0) if res is not None else 0),
         ptr(out), out.stride(0), M, N, K, epi, stream())
    return out


ROWS_MAX = 192   # rows one eagle_gemm_rows_bf16 call takes
_rows_ws_cache = {}


ROWS_COUNTER_BYTES = 16384   # GR_COUNTER_BYTES: the head of the workspace holds the per-tile arrival counters


def _rows_workspace(nbytes: int, device) -> torch.Tensor:
    """Workspace of eagle_gemm_rows_bf16, one per (device, stream).  Only its head (the per-tile arrival counters) must be
    zero on entry, and every call leaves it zero -- so the buffer is allocated uninitialised and its first 16 KiB are
    zeroed once, when it is (re)allocated.  Under stream capture nothing is allocated or zeroed: the capture reuses the
    device's most recent workspace (the warm-up's; the capture stream is then its only user) and raises if there is none
    that is large enough -- warm the call up before capturing it."""
    key = (device, torch.cuda.current_stream().cuda_stream)
    w = _rows_ws_cache.get(key)
    if w is not None and w.numel() >= nbytes:
        return w
    if torch.cuda.is_current_stream_capturing():
        w = _rows_ws_cache.get((device, "last"))
        if w is None or w.numel() < nbytes:
            raise _hip.HipKernelError("eagle_gemm_rows_bf16 under stream capture needs its workspace from an earlier (warm-up) call")
        return w
    w = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
    w[:ROWS_COUNTER_BYTES].zero_()
    _rows_ws_cache[key] = _rows_ws_cache[(device, "last")] = w
    return w


def linear_rows(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None,
                res: Optional[torch.Tensor] = None, W2: Optional[torch.Tensor] = None,
                idx0: Optional[torch.Tensor] = None, x1: Optional[torch.Tensor] = None,
                idx1: Optional[torch.Tensor] = None, M: Optional[int] = None,
                out: Optional[torch.Tensor] = None, ind0: Optional[torch.Tensor] = None,
                idx0_bstride: int = 0) -> torch.Tensor:
    """linear_skinny's product for a decode batch of up to 192 rows in ONE weight pass (eagle_gemm_rows_bf16): W straight
    to registers, x through LDS, K split over workgroups where the matrix has few column tiles.  W (and W2) [N, K] with
    any row stride that is a multiple of 8; K (and the cat boundary) multiples of 128."""
    _chk(x, "x")
    packed = isinstance(W, PackedWeight)
    if not packed:
        _chk(W, "W")
    if (W2 is not None) and isinstance(W2, PackedWeight) != packed:
        raise ValueError("W2 must be packed like W")
    N, K = W.shape
    K0 = x.shape[-1]
    if x1 is not None:
        _chk(x1, "x1")
        if K0 + x1.shape[-1] != K:
            raise ValueError(f"cat width {K0}+{x1.shape[-1]} != K={K}")
    elif K0 != K:
        raise ValueError(f"x width {K0} != K={K}")
    if M is None:
        M = idx0.numel() if idx0 is not None else x.shape[0]
    if packed:
        if x.stride(-1) != 1 or (W2 is not None and W2.shape != W.shape):
            raise ValueError("x rows must be contiguous along K; W2 shaped as W")
    elif x.stride(-1) != 1 or W.stride(1) != 1 or (W2 is not None and (W2.stride(1) != 1 or W2.stride(0) != W.stride(0) or W2.shape != W.shape)):
        raise ValueError("x rows, W and W2 must be contiguous along K; W2 shaped and strided as W")
    for i in (idx0, idx1, ind0):
        if i is not None and (i.dtype != torch.int64 or (not i.is_contiguous() and not (i is idx0 and ind0 is not None))):
            raise TypeError("row index arrays must be contiguous int64")
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=x.device)
    epi = (EPI_SILU_MUL if W2 is not None else EPI_LINEAR) | (W_PACKED if packed else 0)
    nbytes = query("eagle_gemm_rows_workspace_bytes", M, N, K, epi)
    ws = _rows_workspace(nbytes, x.device) if nbytes else None
    call("eagle_gemm_rows_bf16", ptr(x), ptr(idx0), x.stride(0) if x.dim() > 1 else K0, K0,
         ptr(ind0), (ind0.numel() if ind0 is not None else 0), idx0_bstride,
         ptr(x1), ptr(idx1), (x1.stride(0) if x1 is not None else 0),
         ptr(W.data if packed else W), ptr(W2.data if (packed and W2 is not None) else W2), (K if packed else W.stride(0)), ptr(bias), ptr(res),
         (res.stride(0) if res is not None else 0),
         ptr(out), out.stride(0), M, N, K, epi, ptr(ws), (ws.numel() if ws is not None else 0), stream())
    return out


def pick_resident_weights(weights, budget_bytes: int):
    """Smallest-first subset of ``weights`` whose bytes fit ``budget_bytes`` (the Infinity Cache share a
    draft forward's small matrices may occupy); groups (tuples) are taken or left whole."""
    groups = [w if isinstance(w, (tuple, list)) else (w,) for w in weights]
    groups.sort(key=lambda g: sum(t.numel() * t.element_size() for t in g))
    out, used = [], 0
    for g in groups:
        b = sum(t.numel() * t.element_size() for t in g)
        if used + b > budget_bytes:
            break
        out.extend(g)
        used += b
    return out


def gemm_nt(A: torch.Tensor, B: torch.Tensor, bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
            out: Optional[torch.Tensor] = None, out_dtype=BF16, accumulate: bool = False, ksplit: int = 1) -> torch.Tensor:
    """C[M,N] = A[M,K] @ B[N,K]^T (+bias) (+res) on the LDS-tiled MFMA kernels.  ksplit > 1: split-K
    (few output tiles, very long K: the lm_head dgrad)."""
    _chk(A, "A"), _chk(B, "B")
    M, K = A.shape
    N = B.shape[0]
    if B.shape[1] != K or A.stride(1) != 1 or B.stride(1) != 1:
        raise ValueError("gemm_nt: A [M,K], B [N,K], both contiguous along K")
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs an output tensor")
        out = torch.empty((M, N), dtype=out_dtype, device=A.device)
    if out.dtype not in (BF16, torch.float32) or out.stride(1) != 1:
        raise TypeError("gemm_nt: C must be bf16 or fp32, row-major")
    if ksplit > 1:
        ws = _workspace(query("eagle_gemm_nt_splitk_workspace_bytes", M, N, ksplit), A.device)
        call("eagle_gemm_nt_splitk_bf16", ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(out), out.stride(0), ptr(bias), ptr(res),
             (res.stride(0) if res is not None else 0), M, N, K, int(out.dtype == torch.float32), int(accumulate), ksplit,
             ptr(ws), stream())
        return out
    call("eagle_gemm_nt_bf16", ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(out), out.stride(0), ptr(bias), ptr(res),
         (res.stride(0) if res is not None else 0), M, N, K, int(out.dtype == torch.float32), int(accumulate), stream())
    return out


def gemm_nt_tail_plan(M: int, N: int, K: int, n_cu: int = 256):
    """(N1, ksplit) when a big NT GEMM is worth cutting into whole rounds of 256x256 tiles + a split-K tail, else None.
    The 256x256 kernel runs one workgroup per CU: tiles / n_cu rounds, and a last round with few tiles leaves most CUs idle
    for a whole tile time (lm_head forward, 4096 x 151936: 9504 tiles = 37 rounds + 32 tiles).  Those tail tiles' columns
    go to a second launch cut ksplit ways along K (as many workgroups as CUs, 1/ksplit of a tile time) + the slab sum."""
    tm, tn = (M + 255) // 256, (N + 255) // 256
    total = tm * tn
    tail = total % n_cu
    if M % 256 or K % 512 or K < 2048 or total < n_cu or tail == 0 or tail > n_cu // 2 or tail % tm:
        return None
    ks = min(8, n_cu // tail, K // 512)
    if ks < 2:
        return None
    return (tn - tail // tm) * 256, ks


def gemm_nt_rounds(A: torch.Tensor, B: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """gemm_nt for the very wide products of the training step (the frozen lm_head on 4096 tokens): whole rounds of tiles in
    one launch, the tail columns split along K (gemm_nt_tail_plan).  The tail columns' sums are formed slice by slice (fp32)
    instead of in one pass: same value up to fp32 summation order.  Shapes without a wasteful tail take the plain call."""
    M, K = A.shape
    N = B.shape[0]
    plan = gemm_nt_tail_plan(M, N, K)
    if plan is None:
        return gemm_nt(A, B, out=out)
    N1, ks = plan
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=A.device)
    gemm_nt(A, B[:N1], out=out[:, :N1])
    gemm_nt(A, B[N1:], out=out[:, N1:], ksplit=ks)
    return out


def gemm_tn_ok(A: torch.Tensor, B: torch.Tensor) -> bool:
    """Shapes eagle_gemm_tn_bf16 takes: A [K, M], B [K, N] row-major, M and N multiples of 256, K a multiple of 64."""
    K, M = A.shape
    N = B.shape[1]
    return (B.shape[0] == K and M % 256 == 0 and N % 256 == 0 and K % 64 == 0 and K >= 128 and A.stride(1) == 1 and B.stride(1) == 1
            and A.stride(0) % 8 == 0 and B.stride(0) % 8 == 0 and K * max(A.stride(0), B.stride(0)) * 2 < 2 ** 32
            and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0)


def gemm_tn(A: torch.Tensor, B: torch.Tensor, out: Optional[torch.Tensor] = None, out_dtype=BF16, accumulate: bool = False) -> torch.Tensor:
    """C[M,N] (+)= A^T @ B for K-major operands A [K, M], B [K, N] (weight gradients: no transpose pass)."""
    _chk(A, "A"), _chk(B, "B")
    K, M = A.shape
    N = B.shape[1]
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs an output tensor")
        out = torch.empty((M, N), dtype=out_dtype, device=A.device)
    if out.dtype not in (BF16, torch.float32) or out.stride(1) != 1 or tuple(out.shape) != (M, N):
        raise TypeError("gemm_tn: C must be bf16 or fp32 [M, N], row-major")
    call("eagle_gemm_tn_bf16", ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(out), out.stride(0), M, N, K,
         int(out.dtype == torch.float32), int(accumulate), stream())
    return out


def gemm_nn_ok(A: torch.Tensor, B: torch.Tensor) -> bool:
    """Shapes eagle_gemm_nn_bf16 takes: A [M, K], B [K, N] row-major, N a multiple of 256, K a multiple of 64."""
    M, K = A.shape
    N = B.shape[1]
    return (B.shape[0] == K and N % 256 == 0 and K % 64 == 0 and K >= 128 and A.stride(1) == 1 and B.stride(1) == 1
            and A.stride(0) % 8 == 0 and B.stride(0) % 8 == 0 and K * B.stride(0) * 2 < 2 ** 32 and A.stride(0) <= 2 ** 22
            and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0)


def gemm_nn(A: torch.Tensor, B: torch.Tensor, res: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
            out_dtype=BF16) -> torch.Tensor:
    """C[M,N] = A @ B (+res) for B stored K-major [K, N] (data gradients: the weight as the state_dict stores it)."""
    _chk(A, "A"), _chk(B, "B")
    M, K = A.shape
    N = B.shape[1]
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=A.device)
    if out.dtype not in (BF16, torch.float32) or out.stride(1) != 1 or tuple(out.shape) != (M, N):
        raise TypeError("gemm_nn: C must be bf16 or fp32 [M, N], row-major")
    call("eagle_gemm_nn_bf16", ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(out), out.stride(0), ptr(res),
         (res.stride(0) if res is not None else 0), M, N, K, int(out.dtype == torch.float32), 0, stream())
    return out


def transpose(x: torch.Tensor, out: Optional[torch.Tensor] = None, pad_to: int = 1) -> torch.Tensor:
    """out[C, R_pad] = x[R,C]^T (bf16); R_pad = R rounded up to ``pad_to``, the extra columns zero."""
    _chk(x, "x")
    R, C = x.shape
    if x.stride(1) != 1:
        raise ValueError("transpose: rows must be contiguous")
    Rp = (R + pad_to - 1) // pad_to * pad_to
    if out is None:
        out = torch.empty((C, Rp), dtype=BF16, device=x.device)
    elif tuple(out.shape) != (C, Rp) or out.stride(1) != 1:
        raise ValueError(f"transpose: out must be [{C}, {Rp}] row-major")
    call("eagle_transpose_bf16", ptr(x), x.stride(0), ptr(out), out.stride(0), R, C, Rp, stream())
    return out


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float, want_rstd: bool = False):
    _chk(x, "x"), _chk(w, "w")
    D = x.shape[-1]
    x2 = x.reshape(-1, D)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    y = torch.empty_like(x2)
    rstd = torch.empty(x2.shape[0], dtype=torch.float32, device=x.device) if want_rstd else None
    call("eagle_rmsnorm_fwd_bf16", ptr(x2), ptr(w), ptr(y), ptr(rstd), x2.shape[0], D, float(eps), stream())
    y = y.view(x.shape)
    return (y, rstd) if want_rstd else y


def rmsnorm_bwd(dy, x, w, rstd, res=None, dw: Optional[torch.Tensor] = None, accumulate_dw: bool = False):
    """(dx [N,D] bf16 (+res), dw [D] fp32) of y = w * norm(x)."""
    N, D = x.shape
    dx = torch.empty_like(x)
    if dw is None:
        dw = torch.empty(D, dtype=torch.float32, device=x.device)
        accumulate_dw = False
    ws = _workspace(query("eagle_rmsnorm_bwd_workspace_bytes", N, D), x.device)
    call("eagle_rmsnorm_bwd_bf16", ptr(dy), ptr(x), ptr(w), ptr(rstd), ptr(res), ptr(dx), ptr(dw), int(accumulate_dw),
         ptr(ws), N, D, stream())
    return dx, dw


def embed_concat(emb: torch.Tensor, ids: torch.Tensor, hidden: torch.Tensor) -> torch.Tensor:
    _chk(emb, "emb"), _chk(hidden, "hidden")
    if ids.dtype != torch.int64 or not ids.is_contiguous() or not ids.is_cuda:
        raise TypeError("embed_concat: ids must be a contiguous int64 device tensor")
    if hidden.dim() != 2 or hidden.stride(1) != 1 or ids.numel() != hidden.shape[0]:
        raise ValueError("embed_concat: hidden [N, D] with contiguous rows, ids [N]")
    N, D = hidden.shape
    out = torch.empty((N, 2 * D), dtype=BF16, device=hidden.device)
    call("eagle_embed_concat_bf16", ptr(emb), ptr(ids), ptr(hidden), hidden.stride(0), ptr(out), N, D, emb.shape[0], stream())
    return out


def silu_mul(gu: torch.Tensor) -> torch.Tensor:
    _chk(gu, "gu")
    N, I2 = gu.shape
    act = torch.empty((N, I2 // 2), dtype=BF16, device=gu.device)
    call("eagle_silu_mul_fwd_bf16", ptr(gu), gu.stride(0), ptr(act), act.stride(0), N, I2 // 2, stream())
    return act


def silu_mul_bwd(gu: torch.Tensor, dact: torch.Tensor) -> torch.Tensor:
    N, I2 = gu.shape
    dgu = torch.empty_like(gu)
    call("eagle_silu_mul_bwd_bf16", ptr(gu), gu.stride(0), ptr(dact), dact.stride(0), ptr(dgu), dgu.stride(0), N, I2 // 2, stream())
    return dgu


def colsum(x: torch.Tensor, out: torch.Tensor, accumulate: bool = False) -> torch.Tensor:
    """out[c] (+)= sum_n x[n,c]; out bf16 or fp32 [C]."""
    _chk(x, "x")
    N, C = x.shape
    ws = _workspace(query("eagle_colsum_workspace_bytes", N, C), x.device)
    call("eagle_colsum_bf16", ptr(x), x.stride(0), N, C, ptr(out) if out.dtype == BF16 else None,
         ptr(out) if out.dtype == torch.float32 else None, int(accumulate), ptr(ws), stream())
    return out


def adamw_(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale=None, clip_value=0.0):
    call("eagle_adamw_bf16", ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), param.numel(), float(lr), float(beta1),
         float(beta2), float(eps), float(weight_decay), int(step), ptr(grad_scale), float(clip_value), stream())


def grad_clip_coef(grad: torch.Tensor, max_norm: float) -> torch.Tensor:
    """[coef, norm] fp32 on device: coef = min(1, max_norm / (||grad|| + 1e-6))."""
    out = torch.empty(2, dtype=torch.float32, device=grad.device)
    ws = _workspace(query("eagle_grad_norm_workspace_bytes"), grad.device)
    call("eagle_grad_clip_coef_bf16", ptr(grad), grad.numel(), float(max_norm), ptr(out), ptr(ws), stream())
    return out


def clip_value_(grad: torch.Tensor, clip: float) -> torch.Tensor:
    _chk(grad, "grad")
    call("eagle_clip_value_bf16", ptr(grad), grad.numel(), float(clip), stream())
    return grad


def scale_(x: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """x *= scale[0] in place (scale: fp32 device tensor)."""
    _chk(x, "x"), _chk(scale, "scale", torch.float32)
    call("eagle_scale_bf16", ptr(x), x.numel(), ptr(scale), stream())
    return x


def add_f32_bf16(a: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """bf16(a_f32 + b_bf16), same shape."""
    _chk(a, "a", torch.float32)
    out = torch.empty(a.shape, dtype=BF16, device=a.device)
    call("eagle_add_f32_bf16", ptr(a.contiguous()), ptr(b.contiguous() if b is not None else None), ptr(out), a.numel(), stream())
    return out


def make_eagle_input(hidden: torch.Tensor, ids: torch.Tensor, noise: Optional[torch.Tensor] = None):
    """(hidden_out, target, ids_out) of eagle/train.py:256-269 for hidden [B,L,D], ids [B,L]."""
    _chk(hidden, "hidden")
    B, L, D = hidden.shape
    hidden = hidden.contiguous()
    ho, tg = torch.empty_like(hidden), torch.empty_like(hidden)
    io = torch.empty_like(ids)
    call("eagle_make_eagle_input_bf16", ptr(hidden), ptr(noise.contiguous() if noise is not None else None), ptr(ids.contiguous()),
         ptr(ho), ptr(tg), ptr(io), B, L, D, stream())
    return ho, tg, io


def eagle_input_embed_concat(hidden: torch.Tensor, ids: torch.Tensor, emb: torch.Tensor, low: float = 0.0, high: float = 0.0,
                             seed: int = 0, offset: int = 0, want_noise: bool = False):
    """Fused _make_eagle_input + embedding gather/concat (include/eagle_hip.h: eagle_input_embed_concat_bf16).
    hidden [B, L, D] bf16 (rows contiguous, any batch / row strides), ids [B, >= L] int64 unshifted.
    emb [V, D] bf16, contiguous.  Token ids outside [0, V) are CLAMPED by the kernel (nn.Embedding, eagle/llama2.py:515,
    faults on them instead): a tokenizer / vocabulary mismatch must be caught upstream of this call.
    Returns (x0 [B*L, 2D], target [B, L, D], ids_out [B, L], noise [B, L, D] or None)."""
    _chk(hidden, "hidden"), _chk(emb, "emb")
    B, L, D = hidden.shape
    if emb.dim() != 2 or emb.shape[1] != D or not emb.is_contiguous():
        raise ValueError(f"eagle_input_embed_concat: emb must be a contiguous [V, {D}] table (got {tuple(emb.shape)}, strides {emb.stride()})")
    if hidden.stride(2) != 1 or ids.dtype != torch.int64 or not ids.is_cuda or ids.stride(1) != 1 or ids.shape[0] != B or ids.shape[1] < L:
        raise ValueError("eagle_input_embed_concat: hidden [B, L, D] with contiguous rows, ids [B, >= L] int64 on the device")
    dev = hidden.device
    x0 = torch.empty((B * L, 2 * D), dtype=BF16, device=dev)
    target = torch.empty((B, L, D), dtype=BF16, device=dev)
    ids_out = torch.empty((B, L), dtype=torch.int64, device=dev)
    noise = torch.zeros((B, L, D), dtype=BF16, device=dev) if want_noise else None
    call("eagle_input_embed_concat_bf16", ptr(hidden), hidden.stride(0), hidden.stride(1), ptr(ids), ids.stride(0), ptr(emb),
         emb.shape[0], float(low), float(high), int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), ptr(x0), ptr(target),
         ptr(ids_out), ptr(noise), B, L, D, stream())
    return x0, target, ids_out, noise


def tree_attention(qkv: torch.Tensor, position_ids: Optional[torch.Tensor], cos: torch.Tensor, sin: torch.Tensor,
                   kcache: torch.Tensor, vcache: torch.Tensor, past_len: torch.Tensor, past_off: int,
                   tree_mask: Optional[torch.Tensor], pad_mask: Optional[torch.Tensor], out: torch.Tensor,
                   workspace: torch.Tensor, err_flag: Optional[torch.Tensor], B: int, M: int, H: int, HK: int, hd: int,
                   max_kv: int, causal: bool, rows_per_req: int, row_off: int, pos_off: int = 0,
                   pos_stride: int = 1) -> None:
    """See include/eagle_hip.h: eagle_tree_attention_bf16."""
    _chk(qkv, "qkv"), _chk(kcache, "kcache"), _chk(vcache, "vcache"), _chk(out, "out")
    cap = kcache.shape[2]
    T, tmb = 0, 0
    if tree_mask is not None:
        if tree_mask.dtype != torch.uint8 or not tree_mask.is_contiguous():
            raise TypeError("tree_mask must be contiguous uint8 [M,T] or [B,M,T]")
        T = tree_mask.shape[-1]
        tmb = tree_mask.stride(0) if tree_mask.dim() == 3 else 0
    if pad_mask is not None and (pad_mask.dtype != torch.uint8 or pad_mask.stride(-1) != 1):
        raise TypeError("pad_mask must be uint8 [B, kv]")
    if workspace.numel() * workspace.element_size() < query("eagle_attn_decode_workspace_bytes", B, H, max_kv):
        raise ValueError("attention workspace too small")
    call("eagle_tree_attention_bf16", ptr(qkv), qkv.stride(0), ptr(position_ids), pos_off, pos_stride,
         ptr(cos), ptr(sin), cos.shape[0], ptr(kcache), ptr(vcache), kcache.stride(0), cap, ptr(past_len), past_off,
         ptr(tree_mask), T, tmb, ptr(pad_mask), (pad_mask.stride(0) if pad_mask is not None else 0),
         ptr(out), out.stride(0), ptr(workspace), ptr(err_flag), B, M, H, HK, hd, max_kv, int(causal), rows_per_req,
         row_off, stream())


def rope_qk_(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, L: int, n_heads_qk: int, hd: int,
             position_ids: Optional[torch.Tensor] = None, inverse: bool = False) -> torch.Tensor:
    """In-place RoPE on the q|k sections of qkv [rows, >= n_heads_qk*hd]."""
    _chk(qkv, "qkv")
    call("eagle_rope_qk_bf16", ptr(qkv), qkv.stride(0), ptr(position_ids), ptr(cos), ptr(sin), cos.shape[0],
         qkv.shape[0], L, n_heads_qk, hd, int(inverse), stream())
    return qkv


def attn_train_fwd(qkv: torch.Tensor, B: int, L: int, H: int, hd: int):
    """Causal flash attention over roped qkv [B*L, 3*H*hd] -> (out [B*L, H*hd] bf16, lse [B,H,L] fp32)."""
    _chk(qkv, "qkv")
    out = torch.empty((B * L, H * hd), dtype=BF16, device=qkv.device)
    lse = torch.empty((B, H, L), dtype=torch.float32, device=qkv.device)
    call("eagle_attn_train_fwd_bf16", ptr(qkv), qkv.stride(0), ptr(out), out.stride(0), ptr(lse), B, L, H, hd, stream())
    return out, lse


def attn_train_bwd(qkv: torch.Tensor, out: torch.Tensor, dout: torch.Tensor, lse: torch.Tensor, B: int, L: int, H: int,
                   hd: int, dqkv: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(qkv, "qkv"), _chk(out, "out"), _chk(dout, "dout")
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, H, L), dtype=torch.float32, device=qkv.device)
    call("eagle_attn_train_bwd_bf16", ptr(qkv), qkv.stride(0), ptr(out), out.stride(0), ptr(dout), dout.stride(0),
         ptr(lse), ptr(delta), ptr(dqkv), dqkv.stride(0), B, L, H, hd, stream())
    return dqkv


_ws_cache = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    key = (device, torch.cuda.current_stream().cuda_stream)
    w = _ws_cache.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = w
    return w


def topk(logits: torch.Tensor, k: int, want_values: bool = True,
         out_idx: Optional[torch.Tensor] = None, out_val: Optional[torch.Tensor] = None,
         workspace: Optional[torch.Tensor] = None, block_index: Optional[torch.Tensor] = None,
         rows: Optional[int] = None, lse_out: Optional[torch.Tensor] = None) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
    """(values, indices) of the k largest per row: value desc, index asc.  logits [rows, V].
    block_index (device int64 scalar) + rows: logits is a ring [blocks*rows, V] and block
    ``block_index[0]`` is read (cursor in device memory: graph-capturable)."""
    if logits.dtype == BF16:
        dt = 0
    elif logits.dtype == torch.float32:
        dt = 1
    else:
        raise TypeError("topk: bf16 or fp32 logits")
    if not logits.is_cuda:
        raise _hip.HipKernelError("topk: device tensor required")
    total_rows, V = logits.shape
    topk_limit_check(V, k)
    if block_index is None:
        rows = total_rows
    elif rows is None or total_rows % rows:
        raise ValueError("topk: ring reads need rows per block dividing the ring")
    if logits.stride(1) != 1:
        raise ValueError("topk: logits rows must be contiguous")
    if out_idx is None:
        out_idx = torch.empty((rows, k), dtype=torch.int64, device=logits.device)
    if want_values and out_val is None:
        out_val = torch.empty((rows, k), dtype=logits.dtype, device=logits.device)
    nb = query("eagle_topk_workspace_bytes", rows, V, k)
    ws = workspace if workspace is not None else _workspace(nb, logits.device)
    if ws.numel() * ws.element_size() < nb:
        raise ValueError("topk: workspace too small")
    if lse_out is not None:
        # bf16 only: fp32 log-sum-exp of every row from the same pass (eagle_topk_lse_vocab)
        if dt != 0 or block_index is not None or lse_out.dtype != torch.float32 or lse_out.numel() < rows:
            raise ValueError("topk: lse_out needs bf16 logits, no ring cursor and an fp32 [rows] output")
        call("eagle_topk_lse_vocab", ptr(logits), logits.stride(0), rows, V, k, ptr(out_idx),
             ptr(out_val) if want_values else None, ptr(lse_out), ptr(ws), stream())
        return (out_val if want_values else None), out_idx
    call("eagle_topk_vocab", ptr(logits), logits.stride(0), dt, rows, V, k, ptr(out_idx),
         ptr(out_val) if want_values else None, ptr(ws), ptr(block_index), rows, stream())
    return (out_val if want_values else None), out_idx


def topk_into_table(logits: torch.Tensor, k: int, table_idx: torch.Tensor, table_val: torch.Tensor, row0: int, n: int,
                    workspace: torch.Tensor) -> None:
    """top-k of logits [B*n, V] (request-major rows) written into table[:, row0:row0+n, :] of the [B, R, k] tables."""
    _chk(logits, "logits")
    rows, V = logits.shape
    topk_limit_check(V, k)
    B, R, kk = table_idx.shape
    if kk != k or rows != B * n or not table_idx.is_contiguous() or not table_val.is_contiguous() or table_val.shape != table_idx.shape:
        raise ValueError("topk_into_table: tables [B, R, k] contiguous, logits [B*n, V]")
    if workspace.numel() * workspace.element_size() < query("eagle_topk_workspace_bytes", rows, V, k):
        raise ValueError("topk: workspace too small")
    call("eagle_topk_vocab_grouped", ptr(logits), logits.stride(0), 0, rows, V, k, table_idx.data_ptr() + row0 * k * 8,
         table_val.data_ptr() + row0 * k * 2, ptr(workspace), n, R * k, stream())


def lm_head_topk_rows_ok(M: int, N: int, K: int, k: int) -> bool:
    """True when lm_head_topk_into_table serves the shape (81..128 rows against a tall head; 1..16 rows with K = 4096, packed W)."""
    return query("eagle_lm_head_topk_rows_workspace_bytes", M, N, K, k) > 0


_head_ws_cache = {}


def lm_head_topk_into_table(x: torch.Tensor, W: torch.Tensor, k: int, table_idx: torch.Tensor, table_val: torch.Tensor,
                            row0: int, n: int, idx0: Optional[torch.Tensor] = None, M: Optional[int] = None,
                            workspace: Optional[torch.Tensor] = None) -> None:
    """top-k(x[idx0] @ W^T) of M = B*n rows written into table[:, row0:row0+n, :] of the [B, R, k] tables, without the logits
    (eagle_lm_head_topk_rows_bf16: candidates out of the GEMM's epilogue + the final merge).  W row-major or a PackedWeight;
    1..16 rows (one request's tree step) are served for a PackedWeight with K = 4096 only.  workspace: uint8 buffer of at least
    eagle_lm_head_topk_rows_workspace_bytes (a caller that replays HIP graphs owns it); None: a per-stream cache."""
    _chk(x, "x")
    packed = isinstance(W, PackedWeight)
    if not packed:
        _chk(W, "W")
    N, K = W.shape
    if M is None:
        M = idx0.numel() if idx0 is not None else x.shape[0]
    B, R, kk = table_idx.shape
    if kk != k or M != B * n or row0 < 0 or row0 + n > R or not table_idx.is_contiguous() or not table_val.is_contiguous() or \
            table_val.shape != table_idx.shape:
        raise ValueError("lm_head_topk_into_table: tables [B, R, k] contiguous, M = B*n rows")
    if x.stride(-1) != 1 or x.shape[-1] != K or (not packed and (W.stride(1) != 1 or W.stride(0) != K)):
        raise ValueError("x rows and W must be contiguous along K")
    if idx0 is not None and (idx0.dtype != torch.int64 or not idx0.is_contiguous()):
        raise TypeError("row index arrays must be contiguous int64")
    need = query("eagle_lm_head_topk_rows_workspace_bytes", M, N, K, k)
    if need <= 0:
        raise ValueError("lm_head_topk_into_table: shape not served (see lm_head_topk_rows_ok)")
    if M <= 16 and not packed:
        raise ValueError("lm_head_topk_into_table: 1..16 rows need a PackedWeight")
    if workspace is not None:
        if workspace.dtype != torch.uint8 or not workspace.is_contiguous() or workspace.numel() < need or workspace.device != x.device:
            raise ValueError("lm_head_topk_into_table: workspace too small or of the wrong kind")
        ws = workspace
    else:
        key = (x.device, torch.cuda.current_stream().cuda_stream)
        ws = _head_ws_cache.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=x.device)
            _head_ws_cache[key] = ws
    call("eagle_lm_head_topk_rows_bf16", ptr(x), ptr(idx0), x.stride(0), None, 0, 0, ptr(W.data if packed else W), (0 if packed else K), None, M, N, K, k,
         table_idx.data_ptr() + row0 * k * 8, table_val.data_ptr() + row0 * k * 2, n, R * k, ptr(ws), ws.numel(), stream())


def first_argmax(logits: torch.Tensor) -> torch.Tensor:
    return topk(logits, 1, want_values=False)[1][:, 0]


def tree_layout(paths, topk: int, expand_all: bool):
    """Host-side validation + sizes of a static tree (eagle_tree_layout).  Returns (paths int32 [n, width] CPU tensor,
    meta dict).  Raises ValueError on an invalid tree."""
    import ctypes
    paths = [list(p) for p in paths]
    if not paths or any(len(p) < 1 for p in paths):
        raise ValueError("tree paths must be non-empty")
    width = max(len(p) for p in paths)
    arr = torch.full((len(paths), width), -1, dtype=torch.int32)
    for i, p in enumerate(paths):
        arr[i, :len(p)] = torch.tensor(p, dtype=torch.int32)
    meta = (ctypes.c_int32 * 32)()
    rc = _hip.lib().eagle_tree_layout(arr.data_ptr(), len(paths), width, int(topk), int(expand_all), ctypes.cast(meta, ctypes.c_void_p))
    if rc:
        raise ValueError(f"invalid draft tree (duplicate path, path without its parent, choice outside top-{topk}, "
                         f"more than 63 paths / depth 15, or a depth with nothing to expand): {_hip._ERR.get(rc, rc)}")
    m = list(meta)
    return arr, {"n_nodes": m[0], "max_depth": m[1], "n_steps": m[2], "n_fwd": m[3], "n_flat_rows": m[4], "P": m[5],
                 "W": m[6], "mask_bytes": m[7], "step_n": m[8:8 + m[2]]}


def tree_build(paths, topk: int, expand_all: bool, device):
    """Every a19 / a21 buffer of a static tree, built on the device (eagle_tree_build).  Returns (meta, dict of tensors)."""
    arr, meta = tree_layout(paths, topk, expand_all)
    n, P_, W, nf = meta["n_nodes"], meta["P"], meta["W"], meta["n_fwd"]
    i64, i32, u8 = torch.int64, torch.int32, torch.uint8
    o = {"tree_indices": torch.empty(n, dtype=i64, device=device), "tree_position_ids": torch.empty(n, dtype=i64, device=device),
         "tree_attn_mask": torch.empty((n, n), dtype=u8, device=device), "retrieve_indices": torch.empty((P_, W), dtype=i64, device=device),
         "parent": torch.empty(n, dtype=i32, device=device), "flat_row": torch.empty(n, dtype=i32, device=device),
         "step_tree_indices": torch.empty(max(nf, 1), dtype=i64, device=device), "step_parent_row": torch.empty(max(nf, 1), dtype=i64, device=device),
         "step_repeat_nums": torch.empty(max(nf, 1), dtype=i32, device=device),
         "step_attn_mask": torch.empty(max(meta["mask_bytes"], 1), dtype=u8, device=device)}
    err = torch.zeros(1, dtype=i32, device=device)
    dpaths = arr.to(device)
    call("eagle_tree_build", ptr(dpaths), arr.shape[0], arr.shape[1], int(topk), int(expand_all), ptr(o["tree_indices"]),
         ptr(o["tree_position_ids"]), ptr(o["tree_attn_mask"]), ptr(o["retrieve_indices"]), ptr(o["parent"]), ptr(o["flat_row"]),
         ptr(o["step_tree_indices"]), ptr(o["step_parent_row"]), ptr(o["step_repeat_nums"]), ptr(o["step_attn_mask"]), ptr(err),
         stream())
    if int(err.item()):
        raise _hip.HipKernelError("eagle_tree_build rejected a tree eagle_tree_layout accepted")
    return meta, o


def tree_gather_ids(flat_ids: torch.Tensor, tree_idx: torch.Tensor, B: int = 1, out=None,
                    batch_stride: Optional[int] = None) -> torch.Tensor:
    """out[b][j] = flat_ids[b].flatten()[tree_idx[j]].  ``flat_ids`` may be a view that starts at a
    request's first row; ``batch_stride`` (elements) then gives the distance between requests."""
    n = tree_idx.numel()
    if batch_stride is None:
        flat_ids = flat_ids.reshape(B, -1)
        batch_stride = flat_ids.stride(0)
    if out is None:
        out = torch.empty((B, n), dtype=torch.int64, device=flat_ids.device)
    call("eagle_tree_gather_ids", ptr(flat_ids), batch_stride, ptr(tree_idx), n, ptr(out), B, stream())
    return out


def greedy_accept(root_token: torch.Tensor, flat_ids: torch.Tensor, tree_indices: torch.Tensor,
                  retrieve: torch.Tensor, target_argmax: torch.Tensor, prev_len: Optional[torch.Tensor],
                  B: int = 1, want_debug: bool = False, next_ids: Optional[torch.Tensor] = None,
                  n_acc: Optional[torch.Tensor] = None, result: Optional[torch.Tensor] = None,
                  last_idx: Optional[torch.Tensor] = None, hid_idx0: Optional[torch.Tensor] = None,
                  len_cur: Optional[torch.Tensor] = None, len_nxt: Optional[torch.Tensor] = None,
                  cursor: Optional[torch.Tensor] = None):
    """result [B,4] int64 = {best_path, accept_length, next_token, new_len}."""
    n_nodes = tree_indices.numel()
    P_, W = retrieve.shape
    dev = flat_ids.device
    flat = flat_ids.reshape(B, -1)
    if result is None:
        result = torch.empty((B, 4), dtype=torch.int64, device=dev)
    tok = cart = acc = None
    if want_debug:
        tok = torch.empty((B, n_nodes), dtype=torch.int64, device=dev)
        cart = torch.empty((B, P_, W), dtype=torch.int64, device=dev)
        acc = torch.empty((B, P_), dtype=torch.int32, device=dev)
    call("eagle_greedy_accept", ptr(root_token), ptr(flat), flat.stride(0), ptr(tree_indices), n_nodes,
         ptr(retrieve), P_, W, ptr(target_argmax), ptr(prev_len), ptr(tok), ptr(cart), ptr(acc), ptr(result), ptr(next_ids), ptr(n_acc),
         ptr(last_idx), ptr(hid_idx0), (hid_idx0.numel() // B if hid_idx0 is not None else 0), ptr(len_cur), ptr(len_nxt),
         ptr(cursor), B, stream())
    return (result, tok, cart, acc) if want_debug else result


def kv_compact(cache: torch.Tensor, B: int, inner: int, retrieve: torch.Tensor, result: torch.Tensor,
               prev_len: torch.Tensor, err_flag: Optional[torch.Tensor] = None) -> None:
    """cache [..., cap, hd] contiguous with the leading dims flattening to (outer, B, inner)."""
    _chk(cache, "cache")
    if not cache.is_contiguous():
        raise ValueError("kv_compact: cache must be contiguous")
    cap, hd = cache.shape[-2], cache.shape[-1]
    n_slabs = cache.numel() // (cap * hd)
    call("eagle_kv_compact_bf16", ptr(cache), n_slabs, cap * hd, hd, B, inner, ptr(retrieve), retrieve.shape[1],
         ptr(result), ptr(prev_len), cap, ptr(err_flag), stream())


def gather_accepted(src: torch.Tensor, retrieve: torch.Tensor, result: torch.Tensor, B: int = 1,
                    root_io: Optional[torch.Tensor] = None, prev_len_io: Optional[torch.Tensor] = None,
                    out: Optional[torch.Tensor] = None, src_block: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(src, "src")
    n_nodes, D = src.shape[-2], src.shape[-1]
    W = retrieve.shape[1]
    if out is None:
        out = torch.empty((B, W, D), dtype=BF16, device=src.device)
    call("eagle_gather_accepted_bf16", ptr(src), n_nodes, D, ptr(retrieve), W, ptr(result), ptr(out), ptr(root_io),
         ptr(prev_len_io), ptr(src_block), B, stream())
    return out


# ------------------------------------------------------------------ loss
def ce_stat_bytes() -> int:
    return query("eagle_ce_stat_bytes")


def _rows_d(x, name, N):
    """predict / target reach the loss kernels as bare pointers with row stride D."""
    _chk(x, name)
    if x.dim() != 2 or x.shape[0] != N or not x.is_contiguous():
        raise ValueError(f"{name}: contiguous [N, D] rows required (got shape {tuple(x.shape)}, strides {x.stride()})")


def topk_limit_check(V: int, k: int) -> None:
    """eagle_topk_vocab merges ceil(V/1024)*k stage-1 candidates per row in one 256-thread workgroup (<= 4096)."""
    if not 1 <= k <= 16 or ((V + 1023) // 1024) * k > 4096:
        raise ValueError(f"top-{k} over a {V}-entry vocabulary exceeds the kernel's limits (k <= 16, ceil(V/1024)*k <= 4096)")


def ce_partial(t_logits, o_logits, v0: int = 0, predict=None, target=None):
    _chk(t_logits, "t_logits"), _chk(o_logits, "o_logits")
    N, V = t_logits.shape
    if predict is not None:
        _rows_d(predict, "predict", N), _rows_d(target, "target", N)
    if t_logits.stride() != o_logits.stride() or t_logits.stride(1) != 1:
        raise ValueError("logit tensors must share one row-major layout")
    D = predict.shape[-1] if predict is not None else 0
    stats = torch.empty((N, ce_stat_bytes() // 4), dtype=torch.float32, device=t_logits.device)
    call("eagle_ce_partial_bf16", ptr(t_logits), ptr(o_logits), t_logits.stride(0), N, V, v0,
         ptr(predict), ptr(target), D, ptr(stats), stream())
    return stats


def ce_combine(stats: torch.Tensor, mask: Optional[torch.Tensor]):
    """stats [R,N,8] or [N,8] -> (sums[4], lse_t[N], lse_o[N], rows[N,4])."""
    if stats.dim() == 2:
        stats = stats[None]
    R, N = stats.shape[0], stats.shape[1]
    dev = stats.device
    lse_t = torch.empty(N, dtype=torch.float32, device=dev)
    lse_o = torch.empty(N, dtype=torch.float32, device=dev)
    rows = torch.empty((N, 4), dtype=torch.float32, device=dev)
    sums = torch.empty(4, dtype=torch.float32, device=dev)
    call("eagle_ce_combine", ptr(stats.contiguous()), R, N, ptr(mask), ptr(lse_t), ptr(lse_o), ptr(rows), ptr(sums), stream())
    return sums, lse_t, lse_o, rows


def ce_smoothl1_bwd(t_logits, o_logits, lse_t, lse_o, mask, g_p: float, g_v: float, predict=None, target=None,
                    in_place: bool = False):
    N, V = t_logits.shape
    if predict is not None:
        _rows_d(predict, "predict", N), _rows_d(target, "target", N)
    dlog = o_logits if in_place else torch.empty_like(o_logits)
    dpred = torch.empty_like(predict) if predict is not None else None
    D = predict.shape[-1] if predict is not None else 0
    call("eagle_ce_smoothl1_bwd_bf16", ptr(t_logits), ptr(o_logits), t_logits.stride(0), N, V, ptr(lse_t), ptr(lse_o),
         ptr(mask), float(g_p), float(g_v), ptr(dlog), dlog.stride(0), ptr(predict), ptr(target), D, ptr(dpred), stream())
    return dlog, dpred


def rope_tables(head_dim: int, n_pos: int, base: float = 10000.0, device=None):
    """bf16 cos/sin tables [n_pos, hd] as LlamaRotaryEmbedding builds them (eagle/llama2.py:70-93:
    fp32 angles, half-split layout, cast to the activation dtype on use :100-102).  Built on the
    host in fp32 so the values do not depend on the device's transcendental rounding."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    t = torch.arange(n_pos, dtype=torch.float32)
    fr = torch.outer(t, inv_freq)
    ang = torch.cat((fr, fr), dim=-1)
    return ang.cos().to(BF16).to(device), ang.sin().to(BF16).to(device)


# ------------------------------------------------------------------ dynamic tree (batch 1)
def row_lse(logits: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 log-sum-exp of every row of bf16 logits [rows, V]."""
    _chk(logits, "logits")
    rows, V = logits.shape
    if logits.stride(1) != 1:
        raise ValueError("row_lse: rows must be contiguous")
    if out is None:
        out = torch.empty(rows, dtype=torch.float32, device=logits.device)
    call("eagle_row_lse_bf16", ptr(logits), logits.stride(0), rows, V, ptr(out), stream())
    return out


def dyn_step(step: int, k: int, last: bool, vals, ids, lse, fscore_in, fpool_in, mask_in, root_row, pool_tok, pool_score,
             pool_parent, ftok, fscore_out, fpool_out, parent_row, mask_out, S: int, B: int = 1) -> None:
    call("eagle_dyn_step", step, k, int(last), ptr(vals), ptr(ids), ptr(lse), ptr(fscore_in), ptr(fpool_in), ptr(mask_in),
         ptr(root_row), ptr(pool_tok), ptr(pool_score), ptr(pool_parent), ptr(ftok), ptr(fscore_out), ptr(fpool_out),
         ptr(parent_row), ptr(mask_out), S, B, stream())


def dyn_finalize(k: int, S: int, T: int, pool_tok, pool_score, pool_parent, root_token, tree_tok, tree_parent, tree_pos,
                 tree_mask, sel, B: int = 1) -> None:
    call("eagle_dyn_finalize", k, S, T, ptr(pool_tok), ptr(pool_score), ptr(pool_parent), ptr(root_token), ptr(tree_tok),
         ptr(tree_parent), ptr(tree_pos), ptr(tree_mask), ptr(sel), B, stream())


def dyn_accept(T: int, W: int, tree_tok, tree_parent, target_argmax, prev_len, path, result, next_ids=None, n_acc=None,
               last_idx=None, len_cur=None, len_nxt=None, cursor=None, B: int = 1) -> None:
    call("eagle_dyn_accept", T, W, ptr(tree_tok), ptr(tree_parent), ptr(target_argmax), ptr(prev_len), ptr(path), ptr(result),
         ptr(next_ids), ptr(n_acc), ptr(last_idx), ptr(len_cur), ptr(len_nxt), ptr(cursor), B, stream())
