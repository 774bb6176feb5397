"""Block attention kernels of the USP path, backed by libusp_hip.so.

Mirrors the callables yunchang/kernels/attention.py exposes to the ring schedules:
  * hip_attn_forward  <->  pytorch_attn_forward(op_type="efficient") (attention.py:44-136) /
                           flash_attn_forward (:165-202)            [the `fwd-only` contract]
  * hip_attn_backward <->  flash_attn_backward (:205-250)           [the `bwd-only` contract;
                           pytorch_attn_backward (:138-159) raises in the reference]
and defines the richer internal seam (`BlockBackend`) the ring schedules of THIS package use, in
which the LSE merge (ring/utils.py:10-51) and the fp32 gradient accumulation
(zigzag_ring_flash_attn.py:147-170) are fused into the kernels.
"""
from __future__ import annotations

import torch

from .. import _C


def kernel_operand(t: torch.Tensor) -> torch.Tensor:
    """A tensor autograd hands to a backward as-is may be anything -- `out.sum().backward()` delivers an EXPANDED scalar
    (every stride 0), a slice of a bigger gradient has odd strides.  The kernels want unit head-dim stride and 16-byte
    multiples elsewhere (flash-attn's `maybe_contiguous`, flash_attn_interface.py, does the same for the reference)."""
    if t.is_contiguous() and t.storage_offset() == 0:       # the common case, one C++ call
        return t
    es = t.element_size()
    ok = t.stride(-1) == 1 and all(t.shape[d] == 1 or (t.stride(d) * es) % 16 == 0 for d in range(t.dim() - 1)) \
        and (t.storage_offset() * es) % 16 == 0
    return t if ok else t.contiguous()


KERNEL_HEAD_DIMS = (32, 64, 128)      # what libusp_hip.so instantiates (include/usp_hip.h: anything else is USP_EUNSUPPORTED)


def kernel_head_dim(D: int) -> int:
    """The head dim the kernels run a `D`-dim problem at: the smallest instantiated one that holds it.  Zero-padding the head
    dim changes neither the scores (the pad contributes 0 to every dot product; the softmax scale stays D ** -0.5) nor the first D
    output dims, so head dims such as 40, 72, 80, 96, 112 -- which flash-attn serves (kernels/attention.py:177-202: any
    multiple of 8 up to 256) -- run through the entry points of this package on padded copies, at the padded dim's cost.
    Above 128 there is no kernel (a 256-dim tile does not fit the 256-register budget of the 8-wave shape)."""
    for d in KERNEL_HEAD_DIMS:
        if D <= d:
            return d
    raise NotImplementedError(f"head_dim {D} > {KERNEL_HEAD_DIMS[-1]} is not supported by the HIP attention kernels")


def pad_head_dim(*tensors):
    """Zero-pad the last dim of each tensor to kernel_head_dim (autograd-aware: torch.nn.functional.pad)."""
    D = tensors[0].shape[-1]
    Dp = kernel_head_dim(D)
    if Dp == D:
        return tensors
    return tuple(torch.nn.functional.pad(t, (0, Dp - D)) for t in tensors)


def needs_grad(*tensors) -> bool:
    """Does autograd have to record this call?  (The ring functions skip their autograd.Function otherwise.)"""
    return torch.is_grad_enabled() and any(t.requires_grad for t in tensors)


class HipBlockBackend:
    """The device backend: thin adapter over the C ABI (include/usp_hip.h).  Immutable: `interleave` is fixed
    at construction, so concurrent callers (the autograd thread running one layer's backward while the main
    thread runs another layer's forward) cannot disturb each other's launch mode."""

    name = "hip"

    def __init__(self, interleave: bool = False):
        # True: every flash launch carries USP_LAUNCH_INTERLEAVE (include/usp_hip.h) so that RCCL's kernels can
        # become resident beside it; False: persistent launches (fastest when the GPU does nothing else).
        self.interleave = bool(interleave)
        self._beside = self if interleave else None

    def beside_transfers(self) -> "HipBlockBackend":
        """The same backend for launches that are meant to run beside transfers on other streams (ring
        degree > 1, pipelined Ulysses exchange)."""
        if self._beside is None:
            self._beside = HipBlockBackend(interleave=True)
        return self._beside

    def fwd(self, q, k, v, softmax_scale, causal, lse, out=None, acc=None, merge_in=False,
            final_begin=0, final_end=None, window=None, k_splits=None):
        _C.flash_fwd(q, k, v, softmax_scale, causal, lse, out, acc, merge_in, final_begin, final_end,
                     interleave=self.interleave, window=window, k_splits=k_splits)

    def delta(self, dout, out, delta):
        _C.bwd_delta(dout, out, delta)

    def bwd(self, dout, q, k, v, lse, delta, dq, dk, dv, softmax_scale, causal, accum_dq=False,
            accum_dk=False, accum_dv=False, dq16=None, dk16=None, dv16=None, window=None, only=None):
        _C.flash_bwd(dout, q, k, v, lse, delta, dq, dk, dv, softmax_scale, causal, accum_dq,
                     accum_dk, accum_dv, dq16, dk16, dv16, interleave=self.interleave, window=window, only=only)

    def fwd_packed(self, q, k, v, seq_q, seq_k, max_q, max_k, softmax_scale, causal, lse, out=None,
                   acc=None, merge_in=False, final_begin=0, final_end=2):
        _C.flash_fwd_packed(q, k, v, seq_q, seq_k, max_q, max_k, softmax_scale, causal, lse, out, acc,
                            merge_in, final_begin, final_end, interleave=self.interleave)

    def bwd_packed(self, dout, q, k, v, lse, delta, seq_q, seq_k, max_q, max_k, dq, dk, dv,
                   softmax_scale, causal, accum_dq=False, accum_dk=False, accum_dv=False, dq16=None,
                   dk16=None, dv16=None):
        _C.flash_bwd_packed(dout, q, k, v, lse, delta, seq_q, seq_k, max_q, max_k, dq, dk, dv,
                            softmax_scale, causal, accum_dq, accum_dk, accum_dv, dq16, dk16, dv16,
                            interleave=self.interleave)

    def merge(self, acc, lse, blk_out, blk_lse, first):
        _C.lse_merge(acc, lse, blk_out, blk_lse, first)

    def cast(self, dst16, src32):
        _C.cast_from_f32(dst16, src32)

    def add(self, dst, a, b):
        _C.add_f32(dst, a, b)

    def copy_rows(self, dst, src, row_bytes, sizes, dst_strides, src_strides):
        _C.copy_rows(dst, src, row_bytes, sizes, dst_strides, src_strides)


_BACKEND = HipBlockBackend()


def get_block_backend(beside_transfers: bool = False):
    """The block backend; `beside_transfers=True` asks for launches that leave room for collectives queued on
    other streams (a persistent flash launch holds every CU until it ends: DESIGN.md section 5).  Test
    backends without that notion are returned as they are."""
    be = _BACKEND
    if beside_transfers and hasattr(be, "beside_transfers"):
        return be.beside_transfers()
    return be


def set_block_backend(backend):
    """Replace the block backend.  Exists for the CPU/gloo orchestration tests, which plug the
    CPU oracle in here (tests/oracle_backend.py); the package itself only ever installs
    HipBlockBackend and has no CPU path.  Returns the previous backend."""
    global _BACKEND
    prev, _BACKEND = _BACKEND, backend
    return prev


def _default_scale(q, softmax_scale):
    return q.shape[-1] ** (-0.5) if softmax_scale is None else softmax_scale


def _check_plain(dropout_p, softcap, alibi_slopes):
    # the hot path only ever passes these defaults (zigzag_ring_flash_attn.py:207,
    # hybrid/attn_layer.py:132-147); anything else is outside this package's scope.  (window_size IS served: the
    # block kernels take flash-attn's (left, right) window -- include/usp_hip.h, USP_ATTN_WINDOW.)
    if dropout_p not in (0, 0.0):
        raise NotImplementedError("dropout_p != 0 is not supported by the HIP attention kernel")
    if softcap not in (None, 0, 0.0):
        raise NotImplementedError("softcap is not supported by the HIP attention kernel")
    if alibi_slopes is not None:
        raise NotImplementedError("alibi_slopes is not supported by the HIP attention kernel")


def window_of(window_size):
    """flash-attn's `window_size` -> (left, right) or None when it selects no window ((-1, -1), None)."""
    return _C._window(window_size)


def hip_attn_forward(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1),
                     softcap=None, alibi_slopes=None, return_softmax=False):
    """`fwd-only` contract of the reference selector (kernels/__init__.py:63-65; call sites
    zigzag_ring_flash_attn.py:29-43, ring_flash_attn.py:36-48):
        (block_out (B,Sq,Hq,D) q.dtype, block_lse (B,Hq,Sq) fp32)
    Unlike the TORCH_* wrappers (attention.py:135) the LSE is NOT rounded to q.dtype."""
    _check_plain(dropout_p, softcap, alibi_slopes)
    B, Sq, Hq, D = q.shape
    scale = _default_scale(q, softmax_scale)
    if kernel_head_dim(D) != D:                 # e.g. 96: on zero-padded copies (kernel_head_dim)
        out, lse = hip_attn_forward(*pad_head_dim(q, k, v), softmax_scale=scale, causal=causal, window_size=window_size)
        return out[..., :D].contiguous(), lse
    out = torch.empty((B, Sq, Hq, D), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, Hq, Sq), dtype=torch.float32, device=q.device)
    get_block_backend().fwd(q, k, v, scale, bool(causal), lse, out=out, window=window_of(window_size))
    return out, lse


def hip_attn_backward(dout, q, k, v, out, softmax_lse, block_dq_buffer, block_dk_buffer,
                      block_dv_buffer, dropout_p=0.0, softmax_scale=None, bwd_causal=False,
                      window_size=(-1, -1), softcap=None, alibi_slopes=None, deterministic=False,
                      rng_state=None, *args, **kwargs):
    """`bwd-only` contract (argument order of kernels/attention.py:205-206): writes dq/dk/dv into
    the caller's (possibly sliced, 16-bit) buffers; `out`/`softmax_lse` are the GLOBAL rows' values
    (zigzag_ring_flash_attn.py:115-137)."""
    _check_plain(dropout_p, softcap, alibi_slopes)
    be = get_block_backend()
    B, Sq, Hq, D = q.shape
    dev = q.device
    if kernel_head_dim(D) != D:                 # on zero-padded copies; the first D dims of the gradients are the answer
        pdo, pq, pk, pv, po = pad_head_dim(dout, q, k, v, out)
        g = [torch.empty_like(t) for t in (pq, pk, pv)]
        hip_attn_backward(pdo, pq, pk, pv, po, softmax_lse, g[0], g[1], g[2], dropout_p, _default_scale(q, softmax_scale),
                          bwd_causal, window_size, softcap, alibi_slopes, deterministic, rng_state)
        for dst, src in zip((block_dq_buffer, block_dk_buffer, block_dv_buffer), g):
            dst.copy_(src[..., :D])
        return
    delta = torch.empty((B, Hq, Sq), dtype=torch.float32, device=dev)
    be.delta(dout, out, delta)
    lse = softmax_lse if softmax_lse.dtype == torch.float32 else softmax_lse.float()
    if lse.stride(-1) != 1:
        lse = lse.contiguous()
    # the kernels round the final result to 16 bits in their epilogues (no fp32 round trip + cast)
    def direct(t):
        return t.dim() == 4 and t.stride(3) == 1 and all(st % 4 == 0 for st in t.stride()[:3])
    tgt = [t if direct(t) else torch.empty(t.shape, dtype=t.dtype, device=dev)
           for t in (block_dq_buffer, block_dk_buffer, block_dv_buffer)]
    be.bwd(dout, q, k, v, lse, delta, None, None, None, _default_scale(q, softmax_scale), bool(bwd_causal),
           dq16=tgt[0], dk16=tgt[1], dv16=tgt[2], window=window_of(window_size))
    for t, dst in zip(tgt, (block_dq_buffer, block_dk_buffer, block_dv_buffer)):
        if t is not dst:
            dst.copy_(t)


class _HipAttnFunc(torch.autograd.Function):
    """`fwd-bwd` stage: an autograd-aware single-device attention (what UlyssesAttention would use,
    yunchang/ulysses/attn_layer.py:48,101-113)."""

    @staticmethod
    def forward(ctx, q, k, v, softmax_scale, causal, return_lse, window_size=(-1, -1)):
        scale = _default_scale(q, softmax_scale)
        q, k, v = kernel_operand(q), kernel_operand(k), kernel_operand(v)
        out, lse = hip_attn_forward(q, k, v, softmax_scale=scale, causal=causal, window_size=window_size)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.scale, ctx.causal, ctx.window_size = scale, bool(causal), window_size
        if return_lse:
            ctx.mark_non_differentiable(lse)
            return out, lse
        return out

    @staticmethod
    def backward(ctx, dout, *_):
        q, k, v, out, lse = ctx.saved_tensors
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        hip_attn_backward(kernel_operand(dout), q, k, v, out, lse, dq, dk, dv, 0.0, ctx.scale, ctx.causal,
                          ctx.window_size)
        return dq, dk, dv, None, None, None, None


def hip_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1),
                  softcap=0.0, alibi_slopes=None, deterministic=False, return_attn_probs=False,
                  *args, **kwargs):
    """`fwd-bwd` contract (flash_attn_func's signature): `out`, or `(out, softmax_lse, None)` with
    return_attn_probs (the probabilities themselves are never materialised: dropout is 0)."""
    _check_plain(dropout_p, softcap, alibi_slopes)
    D = q.shape[-1]
    if kernel_head_dim(D) != D:                 # on zero-padded copies (autograd differentiates the pad and the slice)
        res = hip_attn_func(*pad_head_dim(q, k, v), softmax_scale=_default_scale(q, softmax_scale), causal=causal,
                            window_size=window_size, return_attn_probs=return_attn_probs)
        return (res[0][..., :D], res[1], None) if return_attn_probs else res[..., :D]
    if return_attn_probs:
        out, lse = _HipAttnFunc.apply(q, k, v, softmax_scale, causal, True, window_size)
        return out, lse, None
    return _HipAttnFunc.apply(q, k, v, softmax_scale, causal, False, window_size)
