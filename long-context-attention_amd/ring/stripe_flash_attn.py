"""Stripe ring attention: same surface as yunchang/ring/stripe_flash_attn.py.

Layout: token t lives on ring rank t % P (EXTRACT_FUNC_DICT["strip"]), so every ring step is a
causal block and the work is balanced without the zigzag pairing.  With local index i <-> global
token i*P + r, the K/V of ring rank r_k = r - step are visible to query i iff
    i_k <= i       when r_k <= r   (step <= rank:  plain causal square,        stripe:31-47)
    i_k <= i - 1   when r_k >  r   (step >  rank:  q[:, 1:] x k[:, :-1] causal, :48-66; rows 1: only)
Both shapes are what the HIP kernel already does (bottom-right aligned causal on a square block);
the row offset is a pointer offset on q / out / acc / lse.  Differences from the reference are the
ones listed in zigzag_ring_flash_attn.py (fused merge, fp32 in-place gradients, side-stream relay).
"""
import torch
import torch.distributed as dist

from ..kernels import AttnType
from ..kernels.attention import get_block_backend, kernel_head_dim, kernel_operand, pad_head_dim
from .utils import FULL, KVRelay, group_info, final_grads, travel_dkdv
from .zigzag_ring_flash_attn import _check_hot_path_args


def stripe_fwd_step(be, r, P, step, q, kk, vv, softmax_scale, lse, out, acc):
    """One ring step of the forward (pure schedule logic; also driven by single-GPU tests)."""
    S = q.shape[1]
    last = step == P - 1
    if step <= r:
        # row 0 gets its last update at step r, rows 1: at step P-1
        fe = S if last else (1 if step == r else 0)
        be.fwd(q, kk, vv, softmax_scale, True, lse, out, acc, step > 0, 0, fe)
    elif S > 1:
        be.fwd(q[:, 1:], kk[:, :-1], vv[:, :-1], softmax_scale, True, lse[:, :, 1:], out[:, 1:],
               acc[:, 1:], True, 0, S - 1 if last else 0)


def stripe_bwd_block(be, r, P, step, dout, q, kk, vv, lse, delta, softmax_scale, dq_acc, dk_dst, dv_dst):
    if step <= r:                                  # stripe_flash_attn.py:116-136
        be.bwd(dout, q, kk, vv, lse, delta, dq_acc, dk_dst, dv_dst, softmax_scale, True,
               accum_dq=step > 0)
    elif q.shape[1] > 1:                           # :137-160
        be.bwd(dout[:, 1:], q[:, 1:], kk[:, :-1], vv[:, :-1], lse[:, :, 1:], delta[:, :, 1:],
               dq_acc[:, 1:], dk_dst[:, :-1], dv_dst[:, :-1], softmax_scale, True, accum_dq=True)


def stripe_bwd_fold(be, r, step, dk_acc, dv_acc, dk_blk, dv_blk):
    if step <= r:                                  # :176-178
        be.add(dk_acc, dk_acc, dk_blk)
        be.add(dv_acc, dv_acc, dv_blk)
    elif dk_acc.shape[1] > 1:                      # :179-181
        be.add(dk_acc[:, :-1], dk_acc[:, :-1], dk_blk[:, :-1])
        be.add(dv_acc[:, :-1], dv_acc[:, :-1], dv_blk[:, :-1])


def stripe_flash_attn_forward(process_group, q, k, v, softmax_scale, dropout_p=0, causal=True,
                              window_size=(-1, -1), softcap=0.0, alibi_slopes=None, deterministic=False,
                              attn_type: AttnType = AttnType.HIP, overlap=False):
    assert causal, "stripe flash attn only supports causal attention, if not causal, use ring flash attn instead"
    P, r = group_info(dist, process_group)
    be = get_block_backend(beside_transfers=P > 1 or overlap)
    B, S, H, D = q.shape
    dev = q.device
    out = torch.empty((B, S, H, D), dtype=q.dtype, device=dev)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=dev)
    acc = torch.empty((B, S, H, D), dtype=torch.float32, device=dev) if P > 1 else None
    with KVRelay(process_group, k, v) as relay:
        for step in range(P):
            kk, vv = relay.get(step)
            stripe_fwd_step(be, r, P, step, q, kk, vv, softmax_scale, lse, out, acc)
    return out, lse


def stripe_flash_attn_backward(process_group, dout, q, k, v, out, softmax_lse, softmax_scale,
                               dropout_p=0, causal=True, window_size=(-1, -1), softcap=0.0,
                               alibi_slopes=None, deterministic=False, attn_type: AttnType = AttnType.HIP,
                               overlap=False, tail=None):
    assert causal, "stripe flash attn only supports causal attention, if not causal, ring flash attn instead"
    P, r = group_info(dist, process_group)
    be = get_block_backend(beside_transfers=P > 1 or overlap)
    B, S, H, D = q.shape
    dev = q.device
    delta = torch.empty((B, H, S), dtype=torch.float32, device=dev)
    be.delta(dout, out, delta)
    if P == 1:
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        be.bwd(dout, q, k, v, softmax_lse, delta, None, None, None, softmax_scale, True,
               dq16=dq, dk16=dk, dv16=dv)
        return dq, dk, dv
    dq_acc = torch.empty((B, S, H, D), dtype=torch.float32, device=dev)

    def block(step, kk, vv, dk_dst, dv_dst):
        stripe_bwd_block(be, r, P, step, dout, q, kk, vv, softmax_lse, delta, softmax_scale, dq_acc,
                         dk_dst, dv_dst)

    def fold(step, dk_acc, dv_acc, dk_blk, dv_blk):
        stripe_bwd_fold(be, r, step, dk_acc, dv_acc, dk_blk, dv_blk)

    # steps > rank see k[:, :-1] only (:137-160, :179-181); a one-token shard has nothing to do there
    dk_acc, dv_acc = travel_dkdv(process_group, k, v, block, fold, be=be, final_dtype=k.dtype, defer=tail,
                                 extent=lambda rank, step: FULL if step <= rank else (slice(0, S - 1) if S > 1 else None))
    return final_grads(be, (q, k, v), (dq_acc, dk_acc, dv_acc))


class StripeFlashAttnFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes,
                deterministic, return_softmax, group, attn_type):
        if softmax_scale is None:
            softmax_scale = q.shape[-1] ** (-0.5)
        assert alibi_slopes is None
        q, k, v = kernel_operand(q), kernel_operand(k), kernel_operand(v)     # any view a caller holds (maybe_contiguous)
        _check_hot_path_args(dropout_p, window_size, softcap)
        out, softmax_lse = stripe_flash_attn_forward(
            group, q, k, v, softmax_scale=softmax_scale, dropout_p=dropout_p, causal=causal,
            window_size=window_size, softcap=softcap, alibi_slopes=alibi_slopes, deterministic=False,
            attn_type=attn_type)
        ctx.save_for_backward(q, k, v, out, softmax_lse)
        ctx.dropout_p = dropout_p
        ctx.softmax_scale = softmax_scale
        ctx.causal = causal
        ctx.window_size = window_size
        ctx.softcap = softcap
        ctx.alibi_slopes = alibi_slopes
        ctx.deterministic = deterministic
        ctx.group = group
        ctx.attn_type = attn_type
        return out if not return_softmax else (out, softmax_lse, None)

    @staticmethod
    def backward(ctx, dout, *args):
        dout = kernel_operand(dout)
        q, k, v, out, softmax_lse = ctx.saved_tensors
        dq, dk, dv = stripe_flash_attn_backward(
            ctx.group, dout, q, k, v, out, softmax_lse, softmax_scale=ctx.softmax_scale,
            dropout_p=ctx.dropout_p, causal=ctx.causal, window_size=ctx.window_size,
            softcap=ctx.softcap, alibi_slopes=ctx.alibi_slopes, deterministic=ctx.deterministic,
            attn_type=ctx.attn_type)
        return dq, dk, dv, None, None, None, None, None, None, None, None, None, None


def stripe_flash_attn_qkvpacked_func(qkv, dropout_p=0.0, softmax_scale=None, causal=False,
                                     window_size=(-1, -1), softcap=0.0, alibi_slopes=None,
                                     deterministic=False, return_attn_probs=False, group=None,
                                     attn_type: AttnType = AttnType.HIP):
    return StripeFlashAttnFunc.apply(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], dropout_p, softmax_scale,
                                     causal, window_size, softcap, alibi_slopes, deterministic,
                                     return_attn_probs, group, attn_type)


def stripe_flash_attn_kvpacked_func(q, kv, dropout_p=0.0, softmax_scale=None, causal=False,
                                    window_size=(-1, -1), softcap=0.0, alibi_slopes=None,
                                    deterministic=False, return_attn_probs=False, group=None,
                                    attn_type: AttnType = AttnType.HIP):
    return StripeFlashAttnFunc.apply(q, kv[:, :, 0], kv[:, :, 1], dropout_p, softmax_scale, causal,
                                     window_size, softcap, alibi_slopes, deterministic,
                                     return_attn_probs, group, attn_type)


def stripe_flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False,
                           window_size=(-1, -1), softcap=0.0, alibi_slopes=None, deterministic=False,
                           return_attn_probs=False, group=None, attn_type: AttnType = AttnType.HIP,
                           attn_processor=None):
    D = q.shape[-1]
    if kernel_head_dim(D) != D:      # a head dim the kernels do not instantiate (e.g. 96): zero-padded copies
        res = stripe_flash_attn_func(*pad_head_dim(q, k, v), dropout_p, D ** -0.5 if softmax_scale is None else softmax_scale, causal,
                                     window_size, softcap, alibi_slopes, deterministic, return_attn_probs, group, attn_type, attn_processor)
        return (res[0][..., :D],) + tuple(res[1:]) if isinstance(res, tuple) else res[..., :D]
    return StripeFlashAttnFunc.apply(q, k, v, dropout_p, softmax_scale, causal, window_size, softcap,
                                     alibi_slopes, deterministic, return_attn_probs, group, attn_type)
