"""Typed Python wrappers over the C ABI (include/cldrd_hip.h): torch tensors in, device pointers out.

PyTorch is only plumbing here (device memory + the current HIP stream); every wrapper checks dtype / layout /
device on the host (the reference raises Python exceptions before launch) and then calls
the kernel.  There is no fallback: tensors must live on a GPU.
"""
from __future__ import annotations

import torch

import os
import threading

from . import _lib
from ._lib import call

BF16 = torch.bfloat16
F32 = torch.float32
F16 = torch.float16


def _fmt16(t, name):
    """16-bit activation format of a forward operand: bf16 (default) or fp16 (high-precision forward of the query tower)."""
    if t.dtype not in (BF16, F16):
        raise TypeError(f"{name}: expected bf16 or fp16, got {t.dtype}")
    return 1 if t.dtype == F16 else 0


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """Raw handle of torch's current stream on the current device.  torch.cuda.current_stream() builds a Stream object per call
    (~5 us: device index resolution, availability check) - a fifth of the host time of a training step at ~230 calls per step."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _chk(t, dtype, name, dim=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor must be on the GPU (cldrd_amd has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if dim is not None and t.dim() != dim:
        raise ValueError(f"{name}: expected {dim} dims, got {t.dim()}")
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: last dim must be contiguous")
    return t


_TLS = threading.local()      # the library keeps these pointers per thread: so does the mirror that lets the contexts nest


class seed_base:
    """``with seed_base(ptr):`` every launch inside passes its ``seed`` argument as an OFFSET: the kernels add the 64-bit word at the device
    address ``ptr`` at run time (include/cldrd_hip.h: cldrd_set_seed_base).  ``ptr`` None / 0: a no-op context."""

    def __init__(self, ptr):
        self.ptr = int(ptr) if ptr else None

    def __enter__(self):
        if self.ptr:
            self.prev = getattr(_TLS, "seed_base", None)        # contexts nest (two towers): restore, do not clear
            _lib.load().cldrd_set_seed_base(self.ptr)
            _TLS.seed_base = self.ptr

    def __exit__(self, *exc):
        if self.ptr:
            _lib.load().cldrd_set_seed_base(self.prev)
            _TLS.seed_base = self.prev


class loss_scale:
    """``with loss_scale(ptr, interval):`` launches inside apply the loss scale at device address ``ptr`` (the float[72] block of
    ``new_loss_scale_state``; include/cldrd_hip.h: cldrd_set_loss_scale).  ``ptr`` None / 0: a no-op context."""

    def __init__(self, ptr, interval=2000):
        self.ptr, self.interval = (int(ptr) if ptr else None), int(interval)

    def __enter__(self):
        if self.ptr:
            self.prev = getattr(_TLS, "loss_scale", None)
            _lib.load().cldrd_set_loss_scale(self.ptr, self.interval)
            _TLS.loss_scale = (self.ptr, self.interval)

    def __exit__(self, *exc):
        if self.ptr:
            prev = self.prev or (None, 0)
            _lib.load().cldrd_set_loss_scale(prev[0], prev[1])
            _TLS.loss_scale = self.prev


class norm_sink:
    """``with norm_sink(slots) as sink:`` weight-gradient slab reductions and LayerNorm-parameter reductions launched inside also write
    per-workgroup sums of squares of the gradients they write to ``slots`` (fp32, device); ``sink.used`` afterwards = slots written, or -1
    when a launch inside could not contribute (include/cldrd_hip.h: cldrd_set_norm_sink).  ``slots`` None: a no-op context (used = -1)."""

    def __init__(self, slots):
        self.slots, self.used = slots, -1

    def __enter__(self):
        if self.slots is not None:
            _chk(self.slots, F32, "slots", 1)
            _lib.load().cldrd_set_norm_sink(self.slots.data_ptr(), self.slots.numel())
        return self

    def __exit__(self, *exc):
        if self.slots is not None:
            lib = _lib.load()
            self.used = int(lib.cldrd_norm_sink_used())
            lib.cldrd_set_norm_sink(None, 0)


def new_loss_scale_state(device):
    """device float[72]: {S = 1, 1 / S = 1, good steps 0, skipped 0, headroom exponent 0, ..., scratch} (include/cldrd_hip.h)"""
    st = torch.zeros(72, dtype=F32, device=device)
    st[0] = 1.0
    st[1] = 1.0
    return st


def loss_scale_adapt(a, b, state):
    """S from max(|a|, |b|) (fp32 tensors: dL/dCLS of the two towers), written to ``state``; a and b are multiplied by S in place."""
    _chk(state, F32, "state", 1)
    if state.numel() < 72:
        raise ValueError("loss_scale_adapt: state is a float[72] from new_loss_scale_state()")
    for t, n in ((a, "a"), (b, "b")):
        if t is not None:
            _chk(t, F32, n)
            if not t.is_contiguous():
                raise ValueError("loss_scale_adapt: contiguous tensors")
    call("cldrd_loss_scale_adapt", _p(a), a.numel() if a is not None else 0, _p(b), b.numel() if b is not None else 0, _p(state), _stream())


class optim_hyper:
    """``with optim_hyper(ptr):`` adamw_step launches inside read {lr, step size} from the device float[2] at ``ptr``."""

    def __init__(self, ptr):
        self.ptr = int(ptr) if ptr else None

    def __enter__(self):
        if self.ptr:
            self.prev = getattr(_TLS, "optim_hyper", None)
            _lib.load().cldrd_set_optim_hyper(self.ptr)
            _TLS.optim_hyper = self.ptr

    def __exit__(self, *exc):
        if self.ptr:
            _lib.load().cldrd_set_optim_hyper(self.prev)
            _TLS.optim_hyper = self.prev


def write_step_state(seeds, seed0, seed1, hyper, lr, beta1, beta2, adam_step, scale_state=None):
    """seeds: device int64[>= 2]; hyper: device float32[>= 2] (either may be None).  ``scale_state`` (the float[72] loss-scale block,
    optional): Adam's bias-correction exponent becomes ``adam_step`` minus the steps the safety net skipped, read on the device."""
    if scale_state is not None:
        _chk(scale_state, F32, "scale_state", 1)
        if scale_state.numel() < 72:
            raise ValueError("write_step_state: scale_state is the float[72] block of new_loss_scale_state()")
    call("cldrd_write_step_state", _p(seeds), int(seed0) & 0xFFFFFFFFFFFFFFFF, int(seed1) & 0xFFFFFFFFFFFFFFFF, _p(hyper), float(lr), float(beta1),
         float(beta2), int(adam_step), _p(scale_state), _stream())


def copy_segments(dsts, srcs):
    """dst[i] <- src[i] for up to 8 pairs of same-sized contiguous device tensors in one launch (bytes are copied: same dtype expected)."""
    import ctypes as C
    n = len(dsts)
    if n != len(srcs) or not 1 <= n <= 8:
        raise ValueError("copy_segments: 1..8 (dst, src) pairs")
    for d, s_ in zip(dsts, srcs):
        if not (d.is_cuda and s_.is_cuda and d.is_contiguous() and s_.is_contiguous() and d.dtype == s_.dtype and d.numel() == s_.numel()):
            raise ValueError("copy_segments: contiguous device tensors of equal dtype and size")
    src = (C.c_void_p * n)(*[t.data_ptr() for t in srcs])
    dst = (C.c_void_p * n)(*[t.data_ptr() for t in dsts])
    nb = (C.c_size_t * n)(*[t.numel() * t.element_size() for t in srcs])
    call("cldrd_copy_segments", src, dst, nb, n, _stream())


def zero_segments(tensors):
    """up to 8 contiguous device tensors (16-byte aligned, byte sizes multiples of 16) set to zero in one launch (cldrd_zero_segments)"""
    import ctypes as C
    n = len(tensors)
    if not 1 <= n <= 8:
        raise ValueError("zero_segments: 1..8 tensors")
    for t in tensors:
        _chk(t, t.dtype, "zero_segments")
        if not t.is_contiguous():
            raise ValueError("zero_segments: contiguous tensors")
    dst = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
    nb = (C.c_size_t * n)(*[t.numel() * t.element_size() for t in tensors])
    call("cldrd_zero_segments", dst, nb, n, _stream())


def pad_rows(rows: int) -> int:
    """Activation / gradient buffers are allocated with rows rounded up to 64 (allocation granularity only: no kernel reads the
    rows past M any more - the weight-gradient kernel fetches them from a zero page)."""
    return (rows + 63) // 64 * 64


def gemm_nt(A, B, out, M=None, *, bias=None, residual=None, preact=None, gelu_pre=None, act=0, alpha=1.0,
            dropout_p=0.0, seed=0, residual_ln=None, out_copy=None):
    """out[M,N] = epilogue(alpha * A[M,K] @ B[N,K]^T); A, B bf16 (or both fp16: forward flavours, M < 1024); out 16-bit like A, or fp32.

    ``act``: bit 0 = erf-GELU; bit 1 = derivative form: ``preact`` receives gelu'(pre-activation) (act=3), ``gelu_pre`` holds it (act=2).
    ``residual_ln`` = (mean[M], rstd[M], gamma[N], beta[N]): the fp32 ``residual`` is a pre-LN sum and LayerNorm(residual) is what is added.
    fp16 operands with M >= 1024: the forward FFN flavours only (bias + GELU [+ tape]; bias [+ dropout] + residual_ln -> fp32 out).
    ``out_copy`` (bf16, shaped like a 16-bit fp16 ``out``): receives the same values in bf16 - the tape entry for the backward."""
    io_f16 = _fmt16(A, "A")
    dt16 = F16 if io_f16 else BF16
    _chk(A, dt16, "A", 2), _chk(B, dt16, "B", 2)
    if io_f16 and out.dtype == BF16:
        io_f16, dt16 = 3, BF16              # fp16 operands, bf16 result (the QKV projection in front of the bf16 attention kernels)
    M = A.shape[0] if M is None else M
    N, K = B.shape
    if A.shape[1] != K or out.shape[1] != N or out.shape[0] < M or A.shape[0] < M:
        raise ValueError(f"gemm_nt: shape mismatch A{tuple(A.shape)} B{tuple(B.shape)} out{tuple(out.shape)} M={M}")
    out_f32 = 1 if out.dtype == F32 else 0
    if not out_f32:
        _chk(out, dt16, "out", 2)
    if bias is not None:
        _chk(bias, F32, "bias", 1)
    tape16 = False
    for t, n in ((preact, "preact"), (gelu_pre, "gelu_pre")):
        if t is not None:
            if t.dtype == F16:        # the all-fp16 training mode: gelu'(x) lives on the tape in fp16 too
                if io_f16 != 1:
                    raise TypeError(f"gemm_nt: an fp16 {n} goes with fp16 operands and an fp16 / fp32 out")
                tape16 = True
            _chk(t, F16 if tape16 else BF16, n, 2)
    if tape16:
        io_f16 = 5
    res_f32 = 0
    if residual is not None:
        res_f32 = 1 if residual.dtype == F32 else 0        # fp32: the residual stream kept in full precision
        _chk(residual, F32 if res_f32 else BF16, "residual", 2)
        if residual.shape[0] < M or residual.shape[1] != N:
            raise ValueError("gemm_nt: residual must be [>= M, N]")
    mean = rstd = gamma = beta = None
    if residual_ln is not None:
        mean, rstd, gamma, beta = residual_ln
        if not res_f32:
            raise ValueError("gemm_nt: residual_ln needs an fp32 residual (the pre-LN sum)")
        _chk(mean, F32, "ln mean", 1), _chk(rstd, F32, "ln rstd", 1), _chk(gamma, F32, "ln gamma", 1), _chk(beta, F32, "ln beta", 1)
        if mean.numel() < M or rstd.numel() < M or gamma.numel() != N or beta.numel() != N:
            raise ValueError("gemm_nt: residual_ln = (mean[>= M], rstd[>= M], gamma[N], beta[N])")
    if out_copy is not None:
        if not io_f16 or out_f32:
            raise ValueError("gemm_nt: out_copy goes with fp16 operands and a 16-bit out")
        _chk(out_copy, BF16, "out_copy", 2)
        if out_copy.shape[1] != N or out_copy.shape[0] < M or out_copy.stride(0) != out.stride(0):
            raise ValueError("gemm_nt: out_copy must have the layout of out")
    # small-M problems (CLS-only last layer, query tower) are split along K when that fills the chip: fp32 partials in a scratch tensor
    ws, ws_bytes = None, 0
    if M < 1024:
        key = (M, N, K, _TUNING.get("gemm_splitk", 0), _TUNING.get("gemm_nt64", 1))      # the kernel choice is part of the cache key
        ws_bytes = _SPLITK_WS.get(key)
        if ws_bytes is None:
            ws_bytes = _SPLITK_WS[key] = int(_lib.load().cldrd_gemm_nt_splitk_workspace(M, N, K))
        if ws_bytes:
            ws = torch.empty(ws_bytes // 4, dtype=F32, device=A.device)
    call("cldrd_gemm_nt16_ws", _p(A), _p(B), _p(out), M, N, K, A.stride(0), B.stride(0), out.stride(0), _p(bias),
         _p(residual), residual.stride(0) if residual is not None else 0, _p(preact), _p(gelu_pre), act, alpha, dropout_p, seed,
         out_f32, res_f32, io_f16, _p(mean), _p(rstd), _p(gamma), _p(beta), _p(out_copy), _p(ws), ws_bytes, _stream())
    return out


_SPLITK_WS = {}
_TUNING = {}


def set_tuning(key: str, value: int) -> None:
    """cldrd_set_tuning: "gemm_splitk" | "gemm_nt64" | "attn_fwd2" | "attn_bwd2" (include/cldrd_hip.h).  Process-wide; tests use it to reach the
    alternative kernels - the library itself reads no environment variable."""
    call("cldrd_set_tuning", key.encode(), int(value))
    _TUNING[key] = int(value)


def wgrad_workspace_elems(M, N1, N2) -> int:
    return _lib.load().cldrd_wgrad_splits(M, N1, N2) * (N1 * N2 + N1)


def wgrad(dY, X, dW, M, workspace, accumulate=False, dbias=None):
    """dW[N1,N2] (+)= dY[:M]^T @ X[:M] (and dbias[N1] (+)= column sums of dY[:M]); dY, X bf16 with >= M rows; dW, dbias fp32."""
    f16 = dY.dtype == F16
    _chk(dY, F16 if f16 else BF16, "dY", 2), _chk(X, F16 if f16 else BF16, "X", 2), _chk(dW, F32, "dW", 2), _chk(workspace, F32, "workspace")
    N1, N2 = dW.shape
    if dY.shape[1] != N1 or X.shape[1] != N2 or not dW.is_contiguous():
        raise ValueError("wgrad: shape mismatch")
    if dY.shape[0] < M or X.shape[0] < M:
        raise ValueError("wgrad: operands have fewer than M rows")
    if dbias is not None:
        _chk(dbias, F32, "dbias", 1)
    call("cldrd_wgrad16", _p(dY), _p(X), _p(dW), _p(dbias), M, N1, N2, dY.stride(0), X.stride(0), _p(workspace),
         workspace.numel() * 4, (1 if accumulate else 0) | (2 if f16 else 0), _stream())
    return dW


class WgradQueue:
    """Deferred weight gradients of one tower: ``add`` records a problem (and keeps its operands alive), ``flush`` hands everything
    recorded so far to ONE ``cldrd_wgrad_group`` launch on the current stream (include/cldrd_hip.h).  Only the data gradients are
    on the critical path of the backward; grouping the weight gradients removes the token splits, the fp32 slabs and ~50 launches
    per tower and step."""

    def __init__(self):
        self.jobs = []

    def add(self, dY, X, dW, M, dbias=None):
        f16 = dY.dtype == F16        # fp16 operands (the all-fp16 training mode); one format per queue
        _chk(dY, F16 if f16 else BF16, "dY", 2), _chk(X, F16 if f16 else BF16, "X", 2), _chk(dW, F32, "dW", 2)
        if self.jobs and (self.jobs[0][0].dtype == F16) != f16:
            raise TypeError("WgradQueue: fp16 and bf16 problems cannot share a launch")
        N1, N2 = dW.shape
        if dY.shape[1] != N1 or X.shape[1] != N2 or not dW.is_contiguous() or dY.shape[0] < M or X.shape[0] < M:
            raise ValueError("wgrad: shape mismatch")
        if dbias is not None:
            _chk(dbias, F32, "dbias", 1)
        self.jobs.append((dY, X, dW, dbias, int(M)))

    def __len__(self):
        return len(self.jobs)

    def flush(self, accumulate=False):
        jobs, self.jobs = self.jobs, []
        if not jobs:
            return
        import ctypes as C
        n = len(jobs)
        VP, IN = C.c_void_p * n, C.c_int * n
        A, B = VP(*[j[0].data_ptr() for j in jobs]), VP(*[j[1].data_ptr() for j in jobs])
        W, Bi = VP(*[j[2].data_ptr() for j in jobs]), VP(*[(j[3].data_ptr() if j[3] is not None else None) for j in jobs])
        Ms, N1s, N2s = IN(*[j[4] for j in jobs]), IN(*[j[2].shape[0] for j in jobs]), IN(*[j[2].shape[1] for j in jobs])
        lda, ldb = IN(*[j[0].stride(0) for j in jobs]), IN(*[j[1].stride(0) for j in jobs])
        need = _lib.load().cldrd_wgrad_group_workspace(Ms, N1s, N2s, n)
        ws = torch.empty(need, dtype=F32, device=jobs[0][2].device) if need else None
        call("cldrd_wgrad_group", A, B, W, Bi, Ms, N1s, N2s, lda, ldb, n, _p(ws), need * 4,
             (1 if accumulate else 0) | (2 if jobs[0][0].dtype == F16 else 0), _stream())
        # `jobs` held the operands alive until here; they were allocated on the stream this launch is on, so releasing them now
        # is ordered behind it by the caching allocator


def attention_drop_bits(nseq, L, H, dropout_p, device):
    """int32 buffer for the dropout keep bits the forward leaves for the backward at this shape, or None (the backward re-hashes)."""
    n = _lib.load().cldrd_attention_bits_words(nseq, L, H, dropout_p)
    return torch.empty(n, dtype=torch.int32, device=device) if n > 0 else None


def _cu_rows(cu, nseq, mask, what):
    """``cu`` (int32 [nseq + 1] on the device): the PACKED layout - sequence m owns rows cu[m] .. cu[m + 1] of every [Tp, .] tensor, keys beyond
    its length are masked (no mask tensor is read)."""
    _chk(cu, torch.int32, "cu", 1)
    if cu.numel() != nseq + 1 or not cu.is_contiguous():
        raise ValueError(f"{what}: cu must be contiguous int32 [nseq + 1]")
    if mask is not None:
        raise ValueError(f"{what}: a packed batch takes its key mask from cu (pass mask=None)")
    return cu


def _seq_list(seq_list, tile, cu, nseq, L, what):
    """``seq_list`` (int32 on the device) with ``tile``: the launch covers only these sequences of a packed batch, each at most ``tile`` <= L tokens
    long, with the kernels of that tile height (cldrd_attention_*_varlen_list)."""
    if cu is None:
        raise ValueError(f"{what}: a sequence list goes with a packed batch (cu)")
    _chk(seq_list, torch.int32, "seq_list", 1)
    if not seq_list.is_contiguous() or not (0 < seq_list.numel() <= nseq) or not (0 < int(tile) <= L):
        raise ValueError(f"{what}: seq_list must be contiguous int32 with 1 .. nseq entries, 0 < tile <= L")
    return (_p(seq_list), int(seq_list.numel()), int(tile))


def attention_fwd(qkv, mask, ctx, lse, nseq, L, H, dropout_p=0.0, seed=0, drop_bits=None, ctx16=None, full_family=False, cu=None, seq_list=None, tile=0):
    """``ctx16`` (fp16, bf16 pass only): the same context in fp16, for an fp16-operand out-projection; ``ctx`` may then be None.
    ``cu``: qkv / ctx / ctx16 are packed [Tp, .] (see _cu_rows); ``lse`` and ``drop_bits`` keep their padded shapes.  ``seq_list`` / ``tile``:
    see _seq_list."""
    io_f16 = _fmt16(qkv, "qkv")
    entry = "cldrd_attention_fwd_bits"
    tail = ()
    if cu is not None:
        entry, mask_arg = "cldrd_attention_fwd_varlen", _p(_cu_rows(cu, nseq, mask, "attention_fwd"))
        if seq_list is not None:
            entry, tail = "cldrd_attention_fwd_varlen_list", _seq_list(seq_list, tile, cu, nseq, L, "attention_fwd")
    else:
        mask_arg = _p(mask)
        if seq_list is not None:
            raise ValueError("attention_fwd: a sequence list goes with a packed batch (cu)")
    _chk(qkv, F16 if io_f16 else BF16, "qkv", 2)
    if ctx16 is not None:
        if io_f16:
            raise ValueError("attention_fwd: ctx16 goes with a bf16 pass")
        _chk(ctx16, F16, "ctx16", 2)
        if ctx16.shape[1] != H * 64 or not ctx16.is_contiguous():
            raise ValueError("attention: ctx16 must be contiguous [T, H*64]")
        if ctx is None:
            if mask is not None:
                _chk(mask, torch.int64, "mask", 2)
            call(entry, _p(qkv), mask_arg, None, _p(lse), nseq, L, H, dropout_p, seed, 0, _p(drop_bits), _p(ctx16), *tail, _stream())
            return ctx16
    _chk(ctx, F16 if io_f16 else BF16, "ctx", 2)
    if mask is not None:
        _chk(mask, torch.int64, "mask", 2)
        if not mask.is_contiguous() or tuple(mask.shape) != (nseq, L):
            raise ValueError("attention: mask must be contiguous int64 [nseq, L]")
    if qkv.shape[1] != 3 * H * 64 or ctx.shape[1] != H * 64 or not qkv.is_contiguous() or not ctx.is_contiguous():
        raise ValueError("attention: qkv must be [T, 3*H*64], ctx [T, H*64], contiguous")
    if lse is not None:
        _chk(lse, F32, "lse")
    if drop_bits is not None:
        _chk(drop_bits, torch.int32, "drop_bits", 1)
        if drop_bits.numel() != _lib.load().cldrd_attention_bits_words(nseq, L, H, dropout_p):
            raise ValueError("attention_fwd: drop_bits must come from attention_drop_bits() for the same shape")
    if io_f16 and (drop_bits is not None or L > 128 or full_family):
        io_f16 = 5                    # fp16 through the whole kernel family of the bf16 path (persistent kernel, keep bits, L > 128)
    call(entry, _p(qkv), mask_arg, _p(ctx), _p(lse), nseq, L, H, dropout_p, seed, io_f16, _p(drop_bits), _p(ctx16), *tail, _stream())
    return ctx


def attention_bwd(qkv, mask, ctx, dctx, lse, dqkv, nseq, L, H, dropout_p=0.0, seed=0, drop_bits=None, cu=None, seq_list=None, tile=0):
    """``cu``: qkv / ctx / dctx / dqkv are packed [Tp, .] (see _cu_rows); ``seq_list`` / ``tile``: see _seq_list."""
    io_f16 = _fmt16(qkv, "qkv")       # fp16 everywhere (the all-fp16 training mode) or bf16 everywhere
    for t, n in ((qkv, "qkv"), (ctx, "ctx"), (dctx, "dctx"), (dqkv, "dqkv")):
        _chk(t, F16 if io_f16 else BF16, n, 2)
        if not t.is_contiguous():
            raise ValueError(f"attention_bwd: {n} must be contiguous")
    _chk(lse, F32, "lse")
    if drop_bits is not None:
        _chk(drop_bits, torch.int32, "drop_bits", 1)
    if seq_list is not None:
        call("cldrd_attention_bwd_varlen_list", _p(qkv), _p(_cu_rows(cu, nseq, mask, "attention_bwd") if cu is not None else None), _p(ctx), _p(dctx),
             _p(lse), _p(dqkv), nseq, L, H, dropout_p, seed, _p(drop_bits), io_f16, *_seq_list(seq_list, tile, cu, nseq, L, "attention_bwd"), _stream())
        return dqkv
    if cu is not None:
        call("cldrd_attention_bwd_varlen", _p(qkv), _p(_cu_rows(cu, nseq, mask, "attention_bwd")), _p(ctx), _p(dctx), _p(lse), _p(dqkv), nseq, L, H,
             dropout_p, seed, _p(drop_bits), io_f16, _stream())
        return dqkv
    call("cldrd_attention_bwd_x", _p(qkv), _p(mask), _p(ctx), _p(dctx), _p(lse), _p(dqkv), nseq, L, H, dropout_p, seed,
         _p(drop_bits), io_f16, _stream())
    return dqkv


def attention_cls_fwd(qc, kv, mask, ctx, probs, nseq, L, H, dropout_p=0.0, seed=0, ctx16=None, cu=None):
    """``cu``: kv is packed [Tp, 2*H*64] (see _cu_rows); probs stays [nseq, H, L]."""
    io_f16 = _fmt16(qc, "qc")
    dt16 = F16 if io_f16 else BF16
    _chk(qc, dt16, "qc", 2), _chk(kv, dt16, "kv", 2), _chk(probs, F32, "probs")
    if ctx is not None:
        _chk(ctx, dt16, "ctx", 2)
    if ctx16 is not None:
        if io_f16:
            raise ValueError("attention_cls_fwd: ctx16 goes with a bf16 pass")
        _chk(ctx16, F16, "ctx16", 2)
    if ctx is None and ctx16 is None:
        raise ValueError("attention_cls_fwd: no output")
    if kv.shape[1] != 2 * H * 64 or not kv.is_contiguous() or not qc.is_contiguous() or any(t is not None and not t.is_contiguous() for t in (ctx, ctx16)):
        raise ValueError("attention_cls: kv must be contiguous [T, 2*H*64]")
    if cu is not None:
        call("cldrd_attention_cls_fwd_varlen", _p(qc), _p(kv), _p(_cu_rows(cu, nseq, mask, "attention_cls_fwd")), _p(ctx), _p(probs), nseq, L, H,
             dropout_p, seed, io_f16, _p(ctx16), _stream())
        return
    call("cldrd_attention_cls_fwd", _p(qc), _p(kv), _p(mask), _p(ctx), _p(probs), nseq, L, H, dropout_p, seed, io_f16, _p(ctx16), _stream())


def attention_cls_bwd(qc, kv, probs, dctx, dqc, dkv, nseq, L, H, dropout_p=0.0, seed=0, cu=None):
    """``cu``: kv and dkv are packed [Tp, 2*H*64] (see _cu_rows)."""
    io_f16 = _fmt16(qc, "qc")
    for t, n in ((qc, "qc"), (kv, "kv"), (dctx, "dctx"), (dqc, "dqc"), (dkv, "dkv")):
        _chk(t, F16 if io_f16 else BF16, n, 2)
        if not t.is_contiguous():
            raise ValueError(f"attention_cls_bwd: {n} must be contiguous")
    if cu is not None:
        call("cldrd_attention_cls_bwd_varlen", _p(qc), _p(kv), _p(_cu_rows(cu, nseq, None, "attention_cls_bwd")), _p(probs), _p(dctx), _p(dqc), _p(dkv),
             nseq, L, H, dropout_p, seed, io_f16, _stream())
        return
    call("cldrd_attention_cls_bwd_x", _p(qc), _p(kv), _p(probs), _p(dctx), _p(dqc), _p(dkv), nseq, L, H, dropout_p, seed, io_f16, _stream())


def add_rows_strided(dst, src, M, stride_rows):
    """dst[m * stride_rows] += src[m]: bf16 rows (fp32 add, one rounding), or fp32 rows (fp32 gradient stream)."""
    fmt = _stream_fmt(dst)           # fp16 dst (the fp16 gradient stream) takes an fp32 src
    _chk(dst, dst.dtype, "dst", 2), _chk(src, F32 if fmt else BF16, "src", 2)
    call("cldrd_add_rows_strided", _p(dst), _p(src), M, src.shape[1], stride_rows, fmt, _stream())


def ln_partial_elems(T, d) -> int:
    return _lib.load().cldrd_ln_partial_blocks(T) * 3 * d


def embed_ln_fwd(ids, word, pos, type0, gamma, beta, out, mean, rstd, T, L, eps, dropout_p=0.0, seed=0, out32=None, pos_idx=None, out_copy=None):
    """``pos_idx`` (int32 [T], packed batches): the position of every row inside its sequence; None: row % L."""
    _chk(ids, torch.int64, "ids")
    d = word.shape[1]
    if out32 is not None:
        _chk(out32, F32, "out32", 2)
    if pos_idx is not None:
        _chk(pos_idx, torch.int32, "pos_idx", 1)
    call("cldrd_embed_ln_fwd", _p(ids), _p(word), _p(pos), _p(type0), _p(gamma), _p(beta), _p(out), _p(mean), _p(rstd),
         T, L, d, word.shape[0], eps, dropout_p, seed, _p(out32), _fmt16(out, "out"), _p(pos_idx), _p(out_copy), _stream())
    return out


def embed_ln_bwd(dy, ids, word, pos, type0, gamma, mean, rstd, dword, dpos, dtype0, dgamma, dbeta, partial, T, L,
                 dropout_p=0.0, seed=0, accumulate=True, pos_idx=None, dy_branch=None):
    d = word.shape[1]
    if pos_idx is not None:
        _chk(pos_idx, torch.int32, "pos_idx", 1)
    if dy.dtype not in (BF16, F32, F16):
        raise TypeError("embed_ln_bwd: dy must be bf16, fp32 (fp32 gradient stream) or fp16 (fp16 gradient stream)")
    if dy_branch is not None and (dy.dtype not in (F32, F16) or dy_branch.dtype not in (BF16, F16) or dy_branch.shape[1] != dy.shape[1] or dy_branch.shape[0] < T
                                  or (dy.dtype == F16 and dy_branch.dtype != F16)):
        raise ValueError("embed_ln_bwd: dy_branch (bf16 or fp16 [>= T, d]) goes with an fp32 dy, an fp16 one with an fp16 dy")
    call("cldrd_embed_ln_bwd", _p(dy), _p(ids), _p(word), _p(pos), _p(type0), _p(gamma), _p(mean), _p(rstd), _p(dword),
         _p(dpos), _p(dtype0), _p(dgamma), _p(dbeta), _p(partial), T, L, d, word.shape[0], dropout_p, seed,
         1 if accumulate else 0, _p(pos_idx), (1 if dy.dtype == F32 else 0) | (4 if (dy_branch is not None and dy_branch.dtype == F16) else 0)
         | (8 if dy.dtype == F16 else 0), _p(dy_branch), _stream())


# ---- variable-length packing (csrc/pack.hip) ---------------------------------------------------------------------------------------
def unpack_rows16(src_packed, dst_padded, cu, nseq, L):
    """dst[m * L + j] = j < len[m] ? src[cu[m] + j] : 0 for 16-bit rows (cu: int32 [nseq + 1] on the device)."""
    _chk(cu, torch.int32, "cu", 1)
    if src_packed.dtype not in (BF16, F16) or dst_padded.dtype != src_packed.dtype or src_packed.shape[1] != dst_padded.shape[1]:
        raise TypeError("unpack_rows16: 16-bit matrices of the same width")
    if not (src_packed.is_contiguous() and dst_padded.is_contiguous()) or dst_padded.shape[0] < nseq * L or cu.numel() < nseq + 1:
        raise ValueError("unpack_rows16: contiguous operands, dst with nseq * L rows, cu with nseq + 1 entries")
    call("cldrd_unpack_rows16", _p(src_packed), _p(dst_padded), _p(cu), nseq, L, src_packed.shape[1], _stream())
    return dst_padded


def gather_rows(src, idx, dst, n=None):
    """dst[p] = src[idx[p]] for p < n (rows of any dtype whose byte length is a multiple of 16)."""
    _chk(idx, torch.int32, "idx", 1)
    n = idx.numel() if n is None else n
    rb = src.shape[1] * src.element_size()
    if src.dtype != dst.dtype or src.shape[1] != dst.shape[1] or not (src.is_contiguous() and dst.is_contiguous()) or dst.shape[0] < n or idx.numel() < n:
        raise ValueError("gather_rows: shape mismatch")
    if n > 0:
        call("cldrd_gather_rows", _p(src), _p(idx), _p(dst), n, rb, _stream())
    return dst


def gather_i64(src, idx, n=None):
    """src[idx[:n]] for int64 ``src`` (flattened) and int32 ``idx``: the token ids of a packed batch's rows."""
    _chk(src, torch.int64, "src"), _chk(idx, torch.int32, "idx", 1)
    n = idx.numel() if n is None else n
    if not src.is_contiguous() or idx.numel() < n:
        raise ValueError("gather_i64: contiguous src, idx with at least n entries")
    out = torch.empty(n, dtype=torch.int64, device=src.device)
    if n > 0:
        call("cldrd_gather_i64", _p(src), _p(idx), _p(out), n, _stream())
    return out


def _stream_fmt(t):
    """row format code of a gradient-stream tensor: 0 bf16, 1 fp32, 2 fp16"""
    if t.dtype not in (BF16, F32, F16):
        raise TypeError(f"gradient stream tensor: expected bf16, fp32 or fp16, got {t.dtype}")
    return 1 if t.dtype == F32 else (2 if t.dtype == F16 else 0)


def scatter_cls_grad_idx(dcls, g, idx, T):
    _chk(dcls, F32, "dcls", 2), _chk(g, g.dtype, "g", 2), _chk(idx, torch.int32, "idx", 1)
    call("cldrd_scatter_cls_grad_idx", _p(dcls), _p(g), dcls.shape[0], dcls.shape[1], _p(idx), T, _stream_fmt(g), _stream())


def add_rows_idx(dst, src, idx, M):
    """dst[idx[m]] += src[m]: bf16 += bf16, fp32 += fp32, or fp16 dst += fp32 src (the fp16 gradient stream)"""
    fmt = _stream_fmt(dst)
    _chk(dst, dst.dtype, "dst", 2), _chk(src, F32 if fmt else BF16, "src", 2), _chk(idx, torch.int32, "idx", 1)
    call("cldrd_add_rows_idx", _p(dst), _p(src), M, src.shape[1], _p(idx), fmt, _stream())


def layernorm_fwd(x, gamma, beta, out, mean, rstd, T, eps, cls_out=None, cls_stride=0, out32=None, out_copy=None):
    """x bf16, or fp32 (pre-LN sum of the fp32 residual stream; then out32, optional, receives the fp32 output as well).
    ``out_copy`` (bf16, with an fp16 ``out``): the same values in bf16 (the backward's MFMA operand)."""
    x_f32 = 1 if x.dtype == F32 else 0
    out_f16 = _fmt16(out, "out")
    _chk(x, F32 if x_f32 else BF16, "x", 2), _chk(out, F16 if out_f16 else BF16, "out", 2), _chk(gamma, F32, "gamma", 1), _chk(beta, F32, "beta", 1)
    if out32 is not None:
        _chk(out32, F32, "out32", 2)
    d = x.shape[1]
    if out_copy is not None:
        if not out_f16:
            raise ValueError("layernorm_fwd: out_copy goes with an fp16 out")
        _chk(out_copy, BF16, "out_copy", 2)
    call("cldrd_layernorm_fwd", _p(x), _p(gamma), _p(beta), _p(out), _p(mean), _p(rstd), T, d, eps, _p(cls_out),
         cls_stride, x_f32, _p(out32), out_f16, _p(out_copy), _stream())
    return out


class LnReduceQueue:
    """Deferred LayerNorm parameter gradients of one tower: ``layernorm_bwd(..., defer=queue)`` leaves its per-block sums in a
    scratch buffer of its own and records where they go; ``flush`` reduces everything recorded so far in ONE launch
    (include/cldrd_hip.h: cldrd_ln_reduce_group) - bit-identical to the immediate form, one launch instead of one per LayerNorm."""

    def __init__(self):
        self.jobs = []

    def __len__(self):
        return len(self.jobs)

    def flush(self, accumulate=False):
        jobs, self.jobs = self.jobs, []
        if not jobs:
            return
        import ctypes as C
        n = len(jobs)
        VP, IN = C.c_void_p * n, C.c_int * n
        d = jobs[0][5]
        if any(j[5] != d for j in jobs):
            raise ValueError("LnReduceQueue: one width per queue")
        ptr = lambda t: t.data_ptr() if t is not None else None
        call("cldrd_ln_reduce_group", VP(*[j[0].data_ptr() for j in jobs]), IN(*[j[4] for j in jobs]), VP(*[ptr(j[1]) for j in jobs]),
             VP(*[ptr(j[2]) for j in jobs]), VP(*[ptr(j[3]) for j in jobs]), n, d, 1 if accumulate else 0, _stream())
        # `jobs` kept the scratch buffers alive until here (same stream: releasing them now is ordered behind the launch)


def layernorm_bwd(dy, x, mean, rstd, gamma, dx, dx_dropped, dgamma, dbeta, dbias, partial, T, dropout_p=0.0, seed=0,
                  accumulate=True, defer=None, dy_branch=None):
    """``defer`` (an LnReduceQueue): dgamma / dbeta / dbias are not produced by this call but by the queue's next ``flush``;
    ``partial`` is then a buffer of this call's own (ln_partial_elems(T, d) floats) that the queue keeps alive.
    fp32 ``dy`` = the fp32 gradient stream: ``dx`` fp32, ``dx_dropped`` (bf16, required) the MFMA operand copy; ``dy_branch`` (bf16,
    optional) is added to dy on load (the branch's data-gradient GEMM output, instead of a residual add in that GEMM's epilogue)."""
    x_f32 = 1 if x.dtype == F32 else 0
    g_f32 = dy.dtype == F32               # fp32 gradient stream: dy and dx fp32, dx_dropped (the 16-bit MFMA operand) required
    g_f16 = dy.dtype == F16               # fp16 gradient stream (round 5): dy, dx, dy_branch and dx_dropped fp16; dx_dropped optional
    if g_f32:
        if not x_f32 or dx_dropped is None:
            raise ValueError("layernorm_bwd: an fp32 dy needs fp32 x and the bf16 operand copy dx_dropped")
        h16 = dx_dropped.dtype == F16        # the all-fp16 training mode: the operand copy and the branch term are fp16
        _chk(dx, F32, "dx", 2), _chk(dx_dropped, F16 if h16 else BF16, "dx_dropped", 2)
        if dy_branch is not None:
            _chk(dy_branch, F16 if h16 else BF16, "dy_branch", 2)
            if dy_branch.shape[1] != dy.shape[1] or dy_branch.shape[0] < T:
                raise ValueError("layernorm_bwd: dy_branch must be [>= T, d]")
        x_f32 |= 2 | (4 if h16 else 0)
    elif g_f16:
        if not x_f32:
            raise ValueError("layernorm_bwd: an fp16 dy (the fp16 gradient stream) needs fp32 x")
        _chk(dx, F16, "dx", 2)
        if dx_dropped is not None:
            _chk(dx_dropped, F16, "dx_dropped", 2)
        if dy_branch is not None:
            _chk(dy_branch, F16, "dy_branch", 2)
            if dy_branch.shape[1] != dy.shape[1] or dy_branch.shape[0] < T:
                raise ValueError("layernorm_bwd: dy_branch must be [>= T, d]")
        x_f32 |= 4 | 8
    else:
        if dy_branch is not None:
            raise ValueError("layernorm_bwd: dy_branch goes with an fp32 / fp16 dy")
        _chk(dy, BF16, "dy", 2), _chk(dx, BF16, "dx", 2)
    _chk(x, F32 if (x_f32 & 1) else BF16, "x", 2)
    d = x.shape[1]
    if defer is not None:
        for t, nme in ((dgamma, "dgamma"), (dbeta, "dbeta"), (dbias, "dbias")):
            if t is not None:
                _chk(t, F32, nme, 1)
        if partial.numel() < ln_partial_elems(T, d):
            raise ValueError("layernorm_bwd: deferred reduction needs a partial buffer of ln_partial_elems(T, d) floats")
        call("cldrd_layernorm_bwd", _p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dx), _p(dx_dropped), None, None, None,
             _p(partial), T, d, dropout_p, seed, 1 if accumulate else 0, x_f32, _p(dy_branch), _stream())
        defer.jobs.append((partial, dgamma, dbeta, dbias, int(T), int(d)))
        return
    call("cldrd_layernorm_bwd", _p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dx), _p(dx_dropped), _p(dgamma),
         _p(dbeta), _p(dbias), _p(partial), T, d, dropout_p, seed, 1 if accumulate else 0, x_f32, _p(dy_branch), _stream())


def colsum(x, out, partial, T, accumulate=True):
    _chk(x, BF16, "x", 2), _chk(out, F32, "out", 1)
    call("cldrd_colsum_bf16", _p(x), _p(out), _p(partial), T, x.shape[1], x.stride(0), 1 if accumulate else 0, _stream())


def scatter_cls_grad(dcls, g, R, stride, T):
    """g[T, d] (bf16, or fp32 for the fp32 gradient stream) = 0 except rows r * stride <- dcls[r]."""
    _chk(dcls, F32, "dcls", 2), _chk(g, g.dtype, "g", 2)
    call("cldrd_scatter_cls_grad", _p(dcls), _p(g), R, dcls.shape[1], stride, T, _stream_fmt(g), _stream())


def score_fwd(q, p, logits, B, N, mode=0):
    _chk(q, F32, "q", 2), _chk(p, F32, "p", 2), _chk(logits, F32, "logits", 2)
    call("cldrd_score_fwd", _p(q), _p(p), _p(logits), B, N, q.shape[1], mode, _stream())
    return logits


def score_bwd(dlogits, q, p, dq, dp, B, N, mode=0):
    for t, n in ((dlogits, "dlogits"), (q, "q"), (p, "p"), (dq, "dq"), (dp, "dp")):
        _chk(t, F32, n, 2)
    call("cldrd_score_bwd", _p(dlogits), _p(q), _p(p), _p(dq), _p(dp), B, N, q.shape[1], mode, _stream())


LOSS_KINDS = {"kl_div": 0, "margin_mse": 1, "ranknet": 2, "lambda_mrr": 3, "weighted_pointwise": 4}
LAMBDA_SCHEMES = {None: 0, "ndcgLoss1_scheme": 1, "ndcgLoss2_scheme": 2, "lambdaRank_scheme": 3, "ndcgLoss2PP_scheme": 4, "rankNet_scheme": 5,
                  "rankNetWeightedByGTDiff_scheme": 6, "rankNetWeightedByGTDiffPowed_scheme": 7}


def loss_fwd_bwd(kind, y_pred, y_true, *, batch_weight=None, T=1.0, pad_indicator=-1.0, reduction="mean"):
    """Returns (loss_out float[2] on device = {loss, pair count}, grad [B,N])."""
    if reduction not in ("mean", "sum"):
        raise ValueError("Reduction method can be either sum or mean")
    _chk(y_pred, F32, "y_pred", 2), _chk(y_true, F32, "y_true", 2)
    if y_pred.shape != y_true.shape:
        raise ValueError("loss: y_pred and y_true must have the same shape")
    y_pred, y_true = y_pred.contiguous(), y_true.contiguous()
    B, N = y_pred.shape
    out = torch.empty(2, dtype=F32, device=y_pred.device)
    grad = torch.empty_like(y_pred)
    ws = torch.empty(2 * B, dtype=F32, device=y_pred.device)
    if batch_weight is not None:
        _chk(batch_weight, F32, "batch_weight", 1)
    call("cldrd_loss_fwd_bwd", LOSS_KINDS[kind], _p(y_pred), _p(y_true), _p(batch_weight), _p(out), _p(grad), _p(ws), B, N,
         float(T), float(pad_indicator), 1 if reduction == "mean" else 0, _stream())
    return out, grad


def lambda_loss_fwd_bwd(y_pred, y_true, *, eps=1e-4, padded_value_indicator=-1, weighing_scheme=None, k=None, sigma=1.0, mu=10.0,
                        reduction="mean", reduction_log="natural", gain="power"):
    """lambda_loss of reference losses/standard_lambda_rank.py:3 -> (loss_out float[2] = {loss, pairs}, grad [B, N])."""
    if weighing_scheme not in LAMBDA_SCHEMES:
        raise KeyError(weighing_scheme)               # the reference looks the scheme up in globals()
    if gain not in ("power", "linear"):
        raise ValueError(f"{gain} not defined.")
    if reduction_log not in ("natural", "binary"):
        raise ValueError("Reduction logarithm base can be either natural or binary")
    if reduction not in ("mean", "sum"):
        raise ValueError("Reduction method can be either sum or mean")
    _chk(y_pred, F32, "y_pred", 2), _chk(y_true, F32, "y_true", 2)
    if y_pred.shape != y_true.shape:
        raise ValueError("lambda_loss: y_pred and y_true must have the same shape")
    y_pred, y_true = y_pred.contiguous(), y_true.contiguous()
    B, N = y_pred.shape
    out = torch.empty(2, dtype=F32, device=y_pred.device)
    grad = torch.empty_like(y_pred)
    ws = torch.empty(2 * B, dtype=F32, device=y_pred.device)
    call("cldrd_lambda_loss_fwd_bwd", _p(y_pred), _p(y_true), _p(out), _p(grad), _p(ws), B, N, LAMBDA_SCHEMES[weighing_scheme],
         0 if k is None else int(k), float(eps), float(sigma), float(mu), float(padded_value_indicator),
         1 if reduction == "mean" else 0, 1 if reduction_log == "binary" else 0, 1 if gain == "linear" else 0, _stream())
    return out, grad


def logit_norm_reg(logits, reg_lambda, loss_out, grad, reg_out=None):
    """loss_out[0] += reg_lambda * ||logits||_2 and grad += its gradient (reference nway_listwise_1.py:348-350)."""
    _chk(logits, F32, "logits"), _chk(loss_out, F32, "loss_out", 1), _chk(grad, F32, "grad")
    if not logits.is_contiguous() or not grad.is_contiguous() or grad.numel() != logits.numel():
        raise ValueError("logit_norm_reg: logits and grad must be contiguous and of the same size")
    if reg_out is not None:
        _chk(reg_out, F32, "reg_out", 1)
    call("cldrd_logit_norm_reg", _p(logits), logits.numel(), float(reg_lambda), _p(loss_out), _p(grad), _p(reg_out), _stream())


def sqnorm_blocks() -> int:
    return _lib.load().cldrd_sqnorm_blocks()


def sqnorm_partial(g, partial, nblk):
    """partial[0:nblk] <- nblk partial sums of squares of the fp32 vector g (one piece of a norm taken in pieces, see clip_coef)."""
    call("cldrd_sqnorm_partial", _p(g), g.numel(), _p(partial), int(nblk), _stream())


def clip_coef(partial, nblk_total, max_norm, out):
    """out[3] = {norm, min(1, max_norm / (norm + 1e-6)), non-finite flag} from partial[0:nblk_total] (fixed-order fp64 sum)."""
    call("cldrd_clip_coef", _p(partial), int(nblk_total), float(max_norm), _p(out), _stream())


def grad_clip_coef(g, max_norm, partial, out):
    _chk(g, F32, "g", 1)
    call("cldrd_grad_clip_coef", _p(g), g.numel(), float(max_norm), _p(partial), _p(out), _stream())
    return out


def adamw_step(p, g, m, v, decay_flags, shadow, *, lr, beta1, beta2, eps, weight_decay, step, clip=None, shadow16=None, h16_range=None):
    """``shadow16`` (fp16, optional) receives the updated parameters ``[h16_range[0], h16_range[1])`` (``shadow16[0]`` = the first of them)."""
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, F32, n, 1)
    if shadow16 is None:
        call("cldrd_adamw_step", _p(p), _p(g), _p(m), _p(v), _p(decay_flags), _p(shadow), p.numel(), float(lr), float(beta1),
             float(beta2), float(eps), float(weight_decay), int(step), _p(clip), _stream())
        return
    _chk(shadow16, F16, "shadow16", 1)
    lo, hi = int(h16_range[0]), int(h16_range[1])
    if shadow16.numel() < hi - lo:
        raise ValueError("adamw_step: shadow16 is smaller than its range")
    call("cldrd_adamw_step_h16", _p(p), _p(g), _p(m), _p(v), _p(decay_flags), _p(shadow), p.numel(), float(lr), float(beta1),
         float(beta2), float(eps), float(weight_decay), int(step), _p(clip), _p(shadow16), lo, hi, _stream())


def cast_bf16(src, dst):
    _chk(src, F32, "src", 1), _chk(dst, BF16, "dst", 1)
    call("cldrd_cast_bf16", _p(src), _p(dst), src.numel(), _stream())


def transpose_cast_batched(src, dst, desc, tile_prefix, ndesc, total_tiles):
    call("cldrd_transpose_cast_batched", _p(src), _p(dst), _p(desc), _p(tile_prefix), ndesc, total_tiles, _stream())


def transpose_bf16_batched(src, dst, desc, tile_prefix, ndesc, total_tiles):
    """16-bit -> 16-bit transposes: bf16 or fp16 (bytes are moved, nothing is converted)."""
    if src.dtype not in (BF16, F16) or dst.dtype != src.dtype:
        raise TypeError("transpose_bf16_batched: 16-bit tensors of one format")
    _chk(src, src.dtype, "src", 1), _chk(dst, src.dtype, "dst", 1)
    call("cldrd_transpose_bf16_batched", _p(src), _p(dst), _p(desc), _p(tile_prefix), ndesc, total_tiles, _stream())


# ---------------------------------------------------------------------------------------------------- top-k search

def topk_scan_filter(Q, P, thr, counts, cand_rows, cand_scores, tiled=False):
    """Q, P: both fp16 or both bf16 (the MFMA type follows)."""
    dt = Q.dtype
    if dt not in (BF16, F16):
        raise TypeError("topk_scan_filter: Q must be fp16 or bf16")
    _chk(Q, dt, "Q", 2), _chk(P, dt, "P", 2), _chk(thr, F32, "thr", 1)
    _chk(counts, torch.int32, "counts", 1), _chk(cand_rows, torch.int32, "cand_rows", 2), _chk(cand_scores, F32, "cand_scores", 2)
    nq, d = Q.shape
    if P.shape[1] != d or not Q.is_contiguous() or not P.is_contiguous() or counts.numel() < nq + 1:
        raise ValueError("topk_scan_filter: shape mismatch (counts needs nq + 1 entries)")
    call("cldrd_topk_scan_filter_tiled" if tiled else "cldrd_topk_scan_filter", _p(Q), _p(P), nq, P.shape[0], d, _p(thr), _p(counts), _p(cand_rows), _p(cand_scores),
         cand_rows.shape[1], 1 if dt == F16 else 0, _stream())


def cast_f16(src, dst, flag=None):
    """fp32 -> fp16 (RNE); flag (uint32/int32 [1], optional) |= 1 when a value does not fit fp16."""
    _chk(src, F32, "src", 1), _chk(dst, F16, "dst", 1)
    call("cldrd_cast_f16", _p(src), _p(dst), src.numel(), _p(flag), _stream())


def topk_prep_queries(q32, qh, qb, qnorm, flag):
    _chk(q32, F32, "q32", 2), _chk(qh, F16, "qh", 2), _chk(qb, BF16, "qb", 2), _chk(qnorm, F32, "qnorm", 1)
    if not (q32.is_contiguous() and qh.is_contiguous() and qb.is_contiguous()):
        raise ValueError("topk_prep_queries: operands must be contiguous")
    call("cldrd_topk_prep_queries", _p(q32), _p(qh), _p(qb), _p(qnorm), q32.shape[0], q32.shape[1], _p(flag), _stream())


def topk_thresholds(est, qnorm, pmax, d, thr, eps):
    _chk(qnorm, F32, "qnorm", 1), _chk(eps, F32, "eps", 1)
    call("cldrd_topk_thresholds", _p(est), _p(qnorm), float(pmax), int(d), _p(thr), _p(eps), qnorm.numel(), _stream())


def topk_select(counts, cand_rows, cand_scores, kk, thr, eps, rows2, n2, status, khat, exhaustive=False):
    nq = cand_rows.shape[0]
    _chk(counts, torch.int32, "counts", 1), _chk(rows2, torch.int32, "rows2", 2), _chk(n2, torch.int32, "n2", 1), _chk(status, torch.int32, "status", 1)
    if counts.numel() < nq + 1:
        raise ValueError("topk_select: counts needs nq + 1 entries")
    call("cldrd_topk_select", _p(counts), _p(cand_rows), _p(cand_scores), nq, cand_rows.shape[1], int(kk), _p(thr), _p(eps), _p(rows2),
         rows2.shape[1], _p(n2), _p(status), _p(khat), 1 if exhaustive else 0, _stream())


def flatip_search(q32, qh, thr, eps, P16, P32, k, counts, cand_rows, cand_scores, rows2, scores2, n2, status, khat, D, I, exhaustive=False,
                  qtile=128, tiled=False):
    """The whole search of one shard (include/cldrd_hip.h: cldrd_flatip_search); everything device resident, no host sync.
    ``tiled``: scan through the tiled kernels (cannot drop hits; the retry form for passes with status bit 4)."""
    _chk(q32, F32, "q32", 2), _chk(P32, F32, "P32", 2), _chk(thr, F32, "thr", 1), _chk(eps, F32, "eps", 1)
    _chk(D, F32, "D", 2), _chk(I, torch.int32, "I", 2)
    nq, d = q32.shape
    rows = P32.shape[0]
    if not exhaustive:
        _chk(qh, F16, "qh", 2), _chk(P16, F16, "P16", 2)
    if cand_rows.shape[0] < qtile or rows2.shape[0] < qtile:
        raise ValueError("flatip_search: the candidate buffers need qtile rows")
    if counts.numel() < ((nq + qtile - 1) // qtile) * (qtile + 1) or n2.numel() < nq or status.numel() < nq or khat.numel() < nq or D.shape != (nq, k) or I.shape != (nq, k):
        raise ValueError("flatip_search: buffer sizes do not match nq / k")
    if not (q32.is_contiguous() and P32.is_contiguous() and D.is_contiguous() and I.is_contiguous()):
        raise ValueError("flatip_search: operands must be contiguous")
    call("cldrd_flatip_search", _p(q32), _p(qh), _p(thr), _p(eps), _p(P16), _p(P32), rows, d, nq, int(k), int(qtile), _p(counts), _p(cand_rows),
         _p(cand_scores), cand_rows.shape[1], _p(rows2), _p(scores2), rows2.shape[1], _p(n2), _p(status), _p(khat), _p(D), _p(I),
         (1 if exhaustive else 0) | (2 if tiled else 0), _stream())


def topk_kth_largest(scores, S, kth, thr):
    _chk(scores, F32, "scores", 2), _chk(thr, F32, "thr", 1)
    call("cldrd_topk_kth_largest", _p(scores), scores.stride(0), scores.shape[0], S, int(kth), _p(thr), _stream())


def topk_rescore(q32, P32, counts, cand_rows, cand_scores):
    _chk(q32, F32, "q32", 2), _chk(P32, F32, "P32", 2)
    call("cldrd_topk_rescore", _p(q32), _p(P32), q32.shape[1], _p(counts), _p(cand_rows), _p(cand_scores), q32.shape[0],
         cand_rows.shape[1], _stream())


def topk_sort(counts, cand_rows, cand_scores, k, D, I):
    _chk(D, F32, "D", 2), _chk(I, torch.int32, "I", 2)
    call("cldrd_topk_sort", _p(counts), _p(cand_rows), _p(cand_scores), cand_rows.shape[0], cand_rows.shape[1], int(k), _p(D),
         _p(I), _stream())


def row_sqnorm_max(P32) -> float:
    _chk(P32, F32, "P32", 2)
    out = torch.zeros(1, dtype=torch.int32, device=P32.device)
    call("cldrd_row_sqnorm_max", _p(P32), P32.shape[0], P32.shape[1], _p(out), _stream())
    return float(out.view(torch.float32).item())


def gather_cast_rows(src32, dst_bf16, n_out, stride):
    _chk(src32, F32, "src32", 2), _chk(dst_bf16, BF16, "dst_bf16", 2)
    call("cldrd_gather_cast_rows", _p(src32), _p(dst_bf16), n_out, stride, src32.shape[1], _stream())


def index_col_mean(P32):
    """mean row of fp32 [rows, d] (fp64 column sums in a fixed order) -> fp32 [d] (cldrd_index_col_mean)"""
    _chk(P32, F32, "P32", 2)
    rows, d = P32.shape
    mu = torch.empty(d, dtype=F32, device=P32.device)
    nbytes = int(_lib.load().cldrd_index_col_mean_workspace(rows, d))
    ws = torch.empty(nbytes // 8, dtype=torch.float64, device=P32.device)
    call("cldrd_index_col_mean", _p(P32), rows, d, _p(mu), _p(ws), nbytes, _stream())
    return mu


def index_center_cast(P32, mu, P16, sample_bf16, s_stride, s_rows, flag):
    """P16 = fp16(P32 - mu), sample = bf16 of every s_stride-th centred row; returns the DEVICE int32 word holding the bit pattern of
    max_r |P32[r] - mu|^2 as fp32 (cldrd_index_center_cast) - read it with ``.view(torch.float32).item()``"""
    _chk(P32, F32, "P32", 2), _chk(mu, F32, "mu", 1), _chk(P16, F16, "P16", 2)
    if sample_bf16 is not None:
        _chk(sample_bf16, BF16, "sample_bf16", 2)
    cmax = torch.zeros(1, dtype=torch.int32, device=P32.device)
    call("cldrd_index_center_cast", _p(P32), _p(mu), P32.shape[0], P32.shape[1], _p(P16), _p(sample_bf16), int(s_stride), int(s_rows), _p(cmax),
         _p(flag), _stream())
    return cmax


def map_ids(I32, ids_table, id_offset):
    """row positions int32 [...] -> ids int64 [...] on the device: the id table when there is one, else position + id_offset; -1 stays -1"""
    _chk(I32, torch.int32, "I32")
    if ids_table is not None:
        _chk(ids_table, torch.int64, "ids_table", 1)
    out = torch.empty(I32.shape, dtype=torch.int64, device=I32.device)
    call("cldrd_map_ids", _p(I32), _p(ids_table), int(id_offset), _p(out), I32.numel(), _stream())
    return out


# ---------------------------------------------------------------------------------------------------- merge of shard lists

def merge_topk_host(shard_D, shard_I, k, nthreads=0):
    """Host k-way merge (include/cldrd_hip.h: cldrd_merge_topk) of per-shard lists: numpy float32 [nq, k_in] / int64 [nq, k_in] per shard
    -> (D float32 [nq, k], I int64 [nq, k]).  Native host threads; no GPU work."""
    import ctypes as C
    import numpy as np
    world = len(shard_D)
    if world == 0 or len(shard_I) != world:
        raise ValueError("merge_topk_host: one score list and one id list per shard")
    Ds = [np.ascontiguousarray(d, dtype=np.float32) for d in shard_D]
    Is = [np.ascontiguousarray(i, dtype=np.int64) for i in shard_I]
    nq, k_in = Ds[0].shape
    if any(d.shape != (nq, k_in) for d in Ds) or any(i.shape != (nq, k_in) for i in Is):
        raise ValueError("merge_topk_host: every shard list must be [nq, k_in]")
    k = int(k)
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    if nq == 0:
        return D, I
    VP = C.c_void_p * world
    call("cldrd_merge_topk", VP(*[d.ctypes.data for d in Ds]), VP(*[i.ctypes.data for i in Is]), world, nq, k_in, k, D.ctypes.data,
         I.ctypes.data, int(nthreads))
    return D, I


def merge_topk_device(scores, ids, k):
    """Device merge (cldrd_merge_topk_device): scores fp32 [world, nq, k_in], ids int64 [world, nq, k_in] in HBM -> (D fp32 [nq, k],
    I int64 [nq, k]) in HBM; world * k_in <= 8192, k <= world * k_in."""
    _chk(scores, F32, "scores", 3), _chk(ids, torch.int64, "ids", 3)
    if scores.shape != ids.shape or not scores.is_contiguous() or not ids.is_contiguous():
        raise ValueError("merge_topk_device: contiguous [world, nq, k_in] scores and ids")
    world, nq, k_in = scores.shape
    k = int(k)
    D = torch.empty(nq, k, dtype=F32, device=scores.device)
    I = torch.empty(nq, k, dtype=torch.int64, device=scores.device)
    if nq == 0:
        return D, I
    nbytes = int(_lib.load().cldrd_merge_topk_device_workspace(world, nq, k_in, k))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=scores.device)
    call("cldrd_merge_topk_device", _p(scores), _p(ids), world, nq, k_in, k, _p(D), _p(I), _p(ws), nbytes, _stream())
    return D, I
