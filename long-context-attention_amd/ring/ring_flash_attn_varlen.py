"""Basic (contiguous-layout) ring attention over packed variable-length batches: same surface as
yunchang/ring/ring_flash_attn_varlen.py.

q/k/v are (T_local, H, D) token tensors, `cu_seqlens` the LOCAL cumulative lengths; every sequence's
local rows are its contiguous chunk r.  Ring step s sees the K/V of ring rank r-s; under causal only
steps <= r compute and only step 0 is causal (:49-69 forward, :113-158 backward).  MI355X-first
differences as in zigzag_ring_flash_attn_varlen.py (packed-mode kernels, flattened fp32 LSE with the
merge fused into the kernel, fp32 in-place gradient accumulation, K/V relay on a side stream); dq is
returned in q.dtype (the reference hard-codes bfloat16 at :176).
"""
import torch
import torch.distributed as dist

from ..kernels.attention import get_block_backend
from .utils import FULL, KVRelay, group_info, final_grads, travel_dkdv
from .varlen_utils import SeqTables, unflatten_lse
from .zigzag_ring_flash_attn import _check_hot_path_args


def basic_varlen_fwd_step(be, r, P, step, causal, tb: SeqTables, q, kk, vv, softmax_scale, lse, out, acc):
    if causal and step > r:
        return
    last_compute = r if causal else P - 1
    be.fwd_packed(q, kk, vv, tb.full, tb.full, tb.max_full, tb.max_full, softmax_scale,
                  bool(causal and step == 0), lse, out, acc, step > 0, 0, 2 if step == last_compute else 0)


def basic_varlen_bwd_block(be, r, P, step, causal, tb: SeqTables, dout, q, kk, vv, lse, delta,
                           softmax_scale, dq_acc, dk_dst, dv_dst):
    """Returns False when the step computes nothing (:113)."""
    if causal and step > r:
        return False
    be.bwd_packed(dout, q, kk, vv, lse, delta, tb.full, tb.full, tb.max_full, tb.max_full, dq_acc,
                  dk_dst, dv_dst, softmax_scale, bool(causal and step == 0), accum_dq=step > 0)
    return True


def ring_flash_attn_varlen_forward(process_group, q, k, v, cu_seqlens, max_seqlen, softmax_scale,
                                   dropout_p=0, causal=True, window_size=(-1, -1), softcap=0.0,
                                   alibi_slopes=None, deterministic=False):
    """Returns (out (T,H,D), lse (H,T) fp32)."""
    P, r = group_info(dist, process_group)
    be = get_block_backend(beside_transfers=P > 1)
    T, H, D = q.shape
    tb = SeqTables(cu_seqlens, max_seqlen, q.device)
    out = torch.empty((T, H, D), dtype=q.dtype, device=q.device)
    lse = torch.empty((H, T), dtype=torch.float32, device=q.device)
    last_compute = r if causal else P - 1
    acc = torch.empty((T, H, D), dtype=torch.float32, device=q.device) if last_compute > 0 else None
    with KVRelay(process_group, k, v) as relay:
        for step in range(P):
            kk, vv = relay.get(step)
            basic_varlen_fwd_step(be, r, P, step, causal, tb, q, kk, vv, softmax_scale, lse, out, acc)
    return out, lse


def ring_flash_attn_varlen_backward(process_group, dout, q, k, v, out, softmax_lse, cu_seqlens,
                                    max_seqlen, softmax_scale, dropout_p=0, causal=True,
                                    window_size=(-1, -1), softcap=0.0, alibi_slopes=None,
                                    deterministic=False):
    P, r = group_info(dist, process_group)
    be = get_block_backend(beside_transfers=P > 1)
    T, H, D = q.shape
    dev, f32 = q.device, torch.float32
    tb = SeqTables(cu_seqlens, max_seqlen, dev)
    dout = dout.contiguous()
    delta = torch.empty((H, T), dtype=f32, device=dev)
    be.delta(dout[None], out[None], delta[None])
    if P == 1:
        # zeros: the kernels do not touch rows outside every sequence's range (padding tokens after cu_seqlens[-1])
        dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
        be.bwd_packed(dout, q, k, v, softmax_lse, delta, tb.full, tb.full, tb.max_full, tb.max_full,
                      None, None, None, softmax_scale, bool(causal), dq16=dq, dk16=dk, dv16=dv)
        return dq, dk, dv
    dq_acc = torch.zeros((T, H, D), dtype=f32, device=dev)

    def block(step, kk, vv, dk_dst, dv_dst):
        return basic_varlen_bwd_block(be, r, P, step, causal, tb, dout, q, kk, vv, softmax_lse, delta,
                                      softmax_scale, dq_acc, dk_dst, dv_dst)

    def fold(step, dk_acc, dv_acc, dk_blk, dv_blk):
        be.add(dk_acc, dk_acc, dk_blk)
        be.add(dv_acc, dv_acc, dv_blk)

    dk_acc, dv_acc = travel_dkdv(process_group, k, v, block, fold, zero=True, be=be, final_dtype=k.dtype,
                                 extent=lambda rank, step: None if (causal and step > rank) else FULL)
    return final_grads(be, (q, k, v), (dq_acc, dk_acc, dv_acc))


class RingFlashAttnVarlenFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, cu_seqlens, max_seqlen, dropout_p, softmax_scale, causal, window_size,
                softcap, alibi_slopes, deterministic, return_softmax, group):
        if softmax_scale is None:
            softmax_scale = q.shape[-1] ** (-0.5)
        assert alibi_slopes is None
        _check_hot_path_args(dropout_p, window_size, softcap)
        k = k.contiguous()
        v = v.contiguous()
        out, lse = ring_flash_attn_varlen_forward(
            group, q, k, v, cu_seqlens, max_seqlen, softmax_scale=softmax_scale, dropout_p=dropout_p,
            causal=causal, window_size=window_size, softcap=softcap, alibi_slopes=alibi_slopes,
            deterministic=False)
        ctx.save_for_backward(q, k, v, out, lse, cu_seqlens)
        ctx.max_seqlen = max_seqlen
        ctx.dropout_p = dropout_p
        ctx.softmax_scale = softmax_scale
        ctx.causal = causal
        ctx.window_size = window_size
        ctx.softcap = softcap
        ctx.alibi_slopes = alibi_slopes
        ctx.deterministic = deterministic
        ctx.group = group
        if not return_softmax:
            return out
        return out, unflatten_lse(lse, cu_seqlens, max_seqlen), None

    @staticmethod
    def backward(ctx, dout, *args):
        q, k, v, out, lse, cu_seqlens = ctx.saved_tensors
        dq, dk, dv = ring_flash_attn_varlen_backward(
            ctx.group, dout, q, k, v, out, lse, cu_seqlens, ctx.max_seqlen,
            softmax_scale=ctx.softmax_scale, dropout_p=ctx.dropout_p, causal=ctx.causal,
            window_size=ctx.window_size, softcap=ctx.softcap, alibi_slopes=ctx.alibi_slopes,
            deterministic=ctx.deterministic)
        return dq, dk, dv, None, None, None, None, None, None, None, None, None, None, None


def ring_flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None,
                                          causal=False, window_size=(-1, -1), softcap=0.0,
                                          alibi_slopes=None, deterministic=False, return_attn_probs=False,
                                          group=None):
    return RingFlashAttnVarlenFunc.apply(qkv[:, 0], qkv[:, 1], qkv[:, 2], cu_seqlens, max_seqlen,
                                         dropout_p, softmax_scale, causal, window_size, softcap,
                                         alibi_slopes, deterministic, return_attn_probs, group)


def ring_flash_attn_varlen_kvpacked_func(q, kv, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None,
                                         causal=False, window_size=(-1, -1), softcap=0.0,
                                         alibi_slopes=None, deterministic=False, return_attn_probs=False,
                                         group=None):
    return RingFlashAttnVarlenFunc.apply(q, kv[:, 0], kv[:, 1], cu_seqlens, max_seqlen, dropout_p,
                                         softmax_scale, causal, window_size, softcap, alibi_slopes,
                                         deterministic, return_attn_probs, group)


def ring_flash_attn_varlen_func(q, k, v, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None,
                                causal=False, window_size=(-1, -1), softcap=0.0, alibi_slopes=None,
                                deterministic=False, return_attn_probs=False, group=None):
    return RingFlashAttnVarlenFunc.apply(q, k, v, cu_seqlens, max_seqlen, dropout_p, softmax_scale,
                                         causal, window_size, softcap, alibi_slopes, deterministic,
                                         return_attn_probs, group)
