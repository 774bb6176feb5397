"""Zigzag ring attention over packed variable-length batches: same surface as
yunchang/ring/zigzag_ring_flash_attn_varlen.py.

q/k/v are (T_local, H, D) token tensors; `cu_seqlens` are the LOCAL cumulative lengths and every
sequence's local rows are [its chunk r | its chunk 2P-1-r].  The step structure is the reference's
(:114-150 forward, :233-281 backward), applied to every sequence at once:
    s == 0 : causal  every sequence x itself
    s <= r : full    every sequence x the FRONT half of the arriving K/V of that sequence
    s >  r : full    the BACK half of every sequence x the arriving K/V of that sequence

MI355X-first differences (results identical up to fp32 rounding order):
  * no gathers: the reference materialises k[half_index0], v[half_index0], q[half_index1],
    dout[half_index1], out[half_index1] with boolean-mask indexing and scatters results back
    (:127-140, :246-264, get_half_lse :45-58); here the kernels address half sequences in place
    through (first_row, rows) tables (include/usp_hip.h, packed mode);
  * the LSE stays flattened (H,T) fp32 from the kernel to the backward -- no flatten / unflatten
    round trips (ring/utils.py:96-117) -- and the merge is fused into the forward kernel's epilogue;
  * fp32 in-place gradient accumulation and the K/V relay on a side stream, as in
    zigzag_ring_flash_attn.py.
`return_attn_probs=True` returns the LSE in the reference's padded (num_seq, H, max_seqlen) layout.
"""
import torch
import torch.distributed as dist

from ..kernels.attention import get_block_backend
from .utils import FULL, KVRelay, group_info, final_grads, travel_dkdv
from .varlen_utils import SeqTables, unflatten_lse
from .zigzag_ring_flash_attn import _check_hot_path_args


def zigzag_varlen_fwd_step(be, r, P, step, tb: SeqTables, q, kk, vv, softmax_scale, lse, out, acc):
    """One ring step of the packed forward on ring rank `r` of `P` (pure schedule logic; final_* count
    half sequences: the front halves are final after step r, the back halves after step P-1)."""
    last = step == P - 1
    if step == 0:                                   # zigzag_ring_flash_attn_varlen.py:120-126
        fe = 2 if last else (1 if r == 0 else 0)
        be.fwd_packed(q, kk, vv, tb.full, tb.full, tb.max_full, tb.max_full, softmax_scale, True, lse,
                      out, acc, False, 0, fe)
    elif step <= r:                                 # :127-135
        fe = 2 if last else (1 if step == r else 0)
        be.fwd_packed(q, kk, vv, tb.full, tb.front, tb.max_full, tb.max_half, softmax_scale, False, lse,
                      out, acc, True, 0, fe)
    else:                                           # :136-144
        be.fwd_packed(q, kk, vv, tb.back, tb.full, tb.max_half, tb.max_full, softmax_scale, False, lse,
                      out, acc, True, 0, 2 if last else 0)


def zigzag_varlen_bwd_block(be, r, P, step, tb: SeqTables, dout, q, kk, vv, lse, delta, softmax_scale,
                            dq_acc, dk_dst, dv_dst):
    """Block backward of ring step `step`; dq accumulates in place (fp32), the dK/dV block goes to
    dk_dst/dv_dst (fp32).  For s <= r only the front-half rows of dk_dst/dv_dst are written."""
    if step == 0:                                   # :233-237
        be.bwd_packed(dout, q, kk, vv, lse, delta, tb.full, tb.full, tb.max_full, tb.max_full, dq_acc,
                      dk_dst, dv_dst, softmax_scale, True)
    elif step <= r:                                 # :239-243
        be.bwd_packed(dout, q, kk, vv, lse, delta, tb.full, tb.front, tb.max_full, tb.max_half, dq_acc,
                      dk_dst, dv_dst, softmax_scale, False, accum_dq=True)
    else:                                           # :244-246
        be.bwd_packed(dout, q, kk, vv, lse, delta, tb.back, tb.full, tb.max_half, tb.max_full, dq_acc,
                      dk_dst, dv_dst, softmax_scale, False, accum_dq=True)


def zigzag_ring_flash_attn_varlen_forward(process_group, q, k, v, cu_seqlens, max_seqlen, softmax_scale,
                                          dropout_p=0, causal=True, window_size=(-1, -1), softcap=0.0,
                                          alibi_slopes=None, deterministic=False):
    """Returns (out (T,H,D), lse (H,T) fp32)."""
    assert causal == True, "zigzag ring is meaningless for causal=False"
    P, r = group_info(dist, process_group)
    be = get_block_backend(beside_transfers=P > 1)
    T, H, D = q.shape
    tb = SeqTables(cu_seqlens, max_seqlen, q.device)
    out = torch.empty((T, H, D), dtype=q.dtype, device=q.device)
    lse = torch.empty((H, T), dtype=torch.float32, device=q.device)
    acc = torch.empty((T, H, D), dtype=torch.float32, device=q.device) if P > 1 else None
    with KVRelay(process_group, k, v) as relay:
        for step in range(P):
            kk, vv = relay.get(step)
            zigzag_varlen_fwd_step(be, r, P, step, tb, q, kk, vv, softmax_scale, lse, out, acc)
    return out, lse


def zigzag_ring_flash_attn_varlen_backward(process_group, dout, q, k, v, out, softmax_lse, cu_seqlens,
                                           max_seqlen, softmax_scale, dropout_p=0, causal=True,
                                           window_size=(-1, -1), softcap=0.0, alibi_slopes=None,
                                           deterministic=False):
    """`softmax_lse` is the flattened (H,T) fp32 LSE of the forward."""
    assert causal == True, "zigzag ring is meaningless for causal=False"
    P, r = group_info(dist, process_group)
    be = get_block_backend(beside_transfers=P > 1)
    T, H, D = q.shape
    dev, f32 = q.device, torch.float32
    tb = SeqTables(cu_seqlens, max_seqlen, dev)
    dout = dout.contiguous()
    delta = torch.empty((H, T), dtype=f32, device=dev)
    be.delta(dout[None], out[None], delta[None])
    if P == 1:   # one block: the kernels round the gradients to q.dtype in their epilogues
        # zeros: the kernels do not touch rows outside every sequence's range (padding tokens after cu_seqlens[-1])
        dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
        be.bwd_packed(dout, q, k, v, softmax_lse, delta, tb.full, tb.full, tb.max_full, tb.max_full,
                      None, None, None, softmax_scale, True, dq16=dq, dk16=dk, dv16=dv)
        return dq, dk, dv
    dq_acc = torch.zeros((T, H, D), dtype=f32, device=dev)

    def block(step, kk, vv, dk_dst, dv_dst):
        if 0 < step <= r:  # only front-half rows are produced: the rest must add as zero (:254-256)
            dk_dst.zero_(); dv_dst.zero_()
        zigzag_varlen_bwd_block(be, r, P, step, tb, dout, q, kk, vv, softmax_lse, delta, softmax_scale,
                                dq_acc, dk_dst, dv_dst)

    def fold(step, dk_acc, dv_acc, dk_blk, dv_blk):
        be.add(dk_acc, dk_acc, dk_blk)
        be.add(dv_acc, dv_acc, dv_blk)

    # the front halves of a packed batch are not one row range: every block travels whole (zeros elsewhere)
    dk_acc, dv_acc = travel_dkdv(process_group, k, v, block, fold, zero=True, be=be, final_dtype=k.dtype,
                                 extent=lambda rank, step: FULL)
    return final_grads(be, (q, k, v), (dq_acc, dk_acc, dv_acc))


class ZigZagRingFlashAttnVarlenFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, cu_seqlens, max_seqlen, dropout_p, softmax_scale, causal, window_size,
                softcap, alibi_slopes, deterministic, return_softmax, group):
        if softmax_scale is None:
            softmax_scale = q.shape[-1] ** (-0.5)
        assert alibi_slopes is None
        _check_hot_path_args(dropout_p, window_size, softcap)
        k = k.contiguous()
        v = v.contiguous()
        out, lse = zigzag_ring_flash_attn_varlen_forward(
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
        dq, dk, dv = zigzag_ring_flash_attn_varlen_backward(
            ctx.group, dout, q, k, v, out, lse, cu_seqlens, ctx.max_seqlen,
            softmax_scale=ctx.softmax_scale, dropout_p=ctx.dropout_p, causal=ctx.causal,
            window_size=ctx.window_size, softcap=ctx.softcap, alibi_slopes=ctx.alibi_slopes,
            deterministic=ctx.deterministic)
        return dq, dk, dv, None, None, None, None, None, None, None, None, None, None, None


def zigzag_ring_flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, dropout_p=0.0,
                                                 softmax_scale=None, causal=False, window_size=(-1, -1),
                                                 softcap=0.0, alibi_slopes=None, deterministic=False,
                                                 return_attn_probs=False, group=None):
    return ZigZagRingFlashAttnVarlenFunc.apply(qkv[:, 0], qkv[:, 1], qkv[:, 2], cu_seqlens, max_seqlen,
                                               dropout_p, softmax_scale, causal, window_size, softcap,
                                               alibi_slopes, deterministic, return_attn_probs, group)


def zigzag_ring_flash_attn_varlen_kvpacked_func(q, kv, cu_seqlens, max_seqlen, dropout_p=0.0,
                                                softmax_scale=None, causal=False, window_size=(-1, -1),
                                                softcap=0.0, alibi_slopes=None, deterministic=False,
                                                return_attn_probs=False, group=None):
    return ZigZagRingFlashAttnVarlenFunc.apply(q, kv[:, 0], kv[:, 1], cu_seqlens, max_seqlen, dropout_p,
                                               softmax_scale, causal, window_size, softcap, alibi_slopes,
                                               deterministic, return_attn_probs, group)


def zigzag_ring_flash_attn_varlen_func(q, k, v, cu_seqlens, max_seqlen, dropout_p=0.0,
                                       softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                                       alibi_slopes=None, deterministic=False, return_attn_probs=False,
                                       group=None):
    return ZigZagRingFlashAttnVarlenFunc.apply(q, k, v, cu_seqlens, max_seqlen, dropout_p, softmax_scale,
                                               causal, window_size, softcap, alibi_slopes, deterministic,
                                               return_attn_probs, group)
