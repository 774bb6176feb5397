"""Basic (contiguous-layout) ring attention: same surface as yunchang/ring/ring_flash_attn.py.

Block structure as the reference (:29-56 forward, :93-143 backward): ring step s sees the K/V of
ring rank r-s; under causal only steps <= r compute and only step 0 is causal.  At ring degree 1
(BASELINE configs C2, C3) this is one kernel call.  The MI355X-first differences are those listed
in zigzag_ring_flash_attn.py (fused merge, fp32 in-place gradient accumulation, K/V relay on a
side stream); additionally dq is returned in q.dtype (the reference hard-codes bfloat16 at :147).
"""
import torch
import torch.distributed as dist

from ..kernels import AttnType
from ..kernels.attention import get_block_backend, kernel_head_dim, kernel_operand, needs_grad, pad_head_dim, window_of
from .utils import FULL, KVRelay, group_info, final_grads, travel_dkdv
from .zigzag_ring_flash_attn import _check_hot_path_args



def _ring_window(window_size, P):
    """flash-attn's window_size -> (left, right) | None.  A window is served where the ring has ONE block (ring degree
    1: the Ulysses-only layouts, the single-GPU path); across ring steps every block would need its own shifted bounds
    (the reference hands the same `window_size` to every block, kernels/attention.py:165-202 -- correct at ring degree
    1 only) and this package refuses instead of computing something else."""
    win = window_of(window_size)
    if win is not None and P > 1:
        raise NotImplementedError("sliding-window attention across ring steps (ring degree > 1) is not supported")
    return win


def _window_kw(win):
    return {} if win is None else {"window": win}


def basic_fwd_step(be, r, P, step, causal, q, kk, vv, softmax_scale, lse, out, acc):
    """One step of the contiguous-layout ring forward (ring_flash_attn.py:29-56); pure schedule
    logic, also driven by the single-GPU tests with virtual ranks."""
    if causal and step > r:
        return
    last_compute = r if causal else P - 1
    fe = q.shape[1] if step == last_compute else 0
    be.fwd(q, kk, vv, softmax_scale, bool(causal and step == 0), lse, out, acc, step > 0, 0, fe)


def basic_bwd_block(be, r, P, step, causal, dout, q, kk, vv, lse, delta, softmax_scale, dq_acc,
                    dk_dst, dv_dst):
    """Block backward of one step (:93-122).  Returns False when the step computes nothing."""
    if causal and step > r:
        return False
    be.bwd(dout, q, kk, vv, lse, delta, dq_acc, dk_dst, dv_dst, softmax_scale,
           bool(causal and step == 0), accum_dq=step > 0)
    return True


def ring_flash_attn_forward(process_group, q, k, v, softmax_scale, dropout_p=0, causal=True,
                            window_size=(-1, -1), softcap=0.0, alibi_slopes=None, deterministic=False,
                            attn_type: AttnType = AttnType.HIP, attn_processor=None, overlap=False):
    P, r = group_info(dist, process_group)
    be = get_block_backend(beside_transfers=P > 1 or overlap)
    B, S, H, D = q.shape
    dev = q.device
    out = torch.empty((B, S, H, D), dtype=q.dtype, device=dev)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=dev)
    win = _ring_window(window_size, P)
    if win is not None:              # ring degree 1: one block, the kernels take flash-attn's window (left, right)
        be.fwd(q, k, v, softmax_scale, bool(causal), lse, out, window=win)
        return out, lse
    last_compute = r if causal else P - 1
    acc = torch.empty((B, S, H, D), dtype=torch.float32, device=dev) if last_compute > 0 else None
    with KVRelay(process_group, k, v) as relay:
        for step in range(P):
            kk, vv = relay.get(step)
            basic_fwd_step(be, r, P, step, causal, q, kk, vv, softmax_scale, lse, out, acc)
    return out, lse


def ring_flash_attn_backward(process_group, dout, q, k, v, out, softmax_lse, softmax_scale,
                             dropout_p=0, causal=True, window_size=(-1, -1), softcap=0.0,
                             alibi_slopes=None, deterministic=False,
                             attn_type: AttnType = AttnType.HIP, overlap=False, tail=None):
    P, r = group_info(dist, process_group)
    be = get_block_backend(beside_transfers=P > 1 or overlap)
    B, S, H, D = q.shape
    dev = q.device
    delta = torch.empty((B, H, S), dtype=torch.float32, device=dev)
    be.delta(dout, out, delta)
    if P == 1:   # one block: the kernels round the gradients to q.dtype in their epilogues
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        be.bwd(dout, q, k, v, softmax_lse, delta, None, None, None, softmax_scale, bool(causal),
               dq16=dq, dk16=dk, dv16=dv, **_window_kw(_ring_window(window_size, P)))
        return dq, dk, dv
    _ring_window(window_size, P)
    dq_acc = torch.empty((B, S, H, D), dtype=torch.float32, device=dev)

    def block(step, kk, vv, dk_dst, dv_dst):
        return basic_bwd_block(be, r, P, step, causal, dout, q, kk, vv, softmax_lse, delta, softmax_scale,
                               dq_acc, dk_dst, dv_dst)

    def fold(step, dk_acc, dv_acc, dk_blk, dv_blk):
        be.add(dk_acc, dk_acc, dk_blk)
        be.add(dv_acc, dv_acc, dv_blk)

    # under causal only steps <= rank compute (:93-122)
    dk_acc, dv_acc = travel_dkdv(process_group, k, v, block, fold, be=be, final_dtype=k.dtype, defer=tail,
                                 extent=lambda rank, step: None if (causal and step > rank) else FULL)
    return final_grads(be, (q, k, v), (dq_acc, dk_acc, dv_acc))


class RingFlashAttnFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes,
                deterministic, return_softmax, group, attn_type, attn_processor):
        if softmax_scale is None:
            softmax_scale = q.shape[-1] ** (-0.5)
        assert alibi_slopes is None
        q, k, v = kernel_operand(q), kernel_operand(k), kernel_operand(v)     # any view a caller holds (maybe_contiguous)
        _check_hot_path_args(dropout_p, (-1, -1), softcap)                    # (the window: ring_flash_attn_forward)
        out, softmax_lse = ring_flash_attn_forward(
            group, q, k, v, softmax_scale=softmax_scale, dropout_p=dropout_p, causal=causal,
            window_size=window_size, softcap=softcap, alibi_slopes=alibi_slopes, deterministic=False,
            attn_type=attn_type, attn_processor=attn_processor)
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
        ctx.attn_processor = attn_processor
        return out if not return_softmax else (out, softmax_lse, None)

    @staticmethod
    def backward(ctx, dout, *args):
        dout = kernel_operand(dout)
        q, k, v, out, softmax_lse = ctx.saved_tensors
        dq, dk, dv = ring_flash_attn_backward(
            ctx.group, dout, q, k, v, out, softmax_lse, softmax_scale=ctx.softmax_scale,
            dropout_p=ctx.dropout_p, causal=ctx.causal, window_size=ctx.window_size,
            softcap=ctx.softcap, alibi_slopes=ctx.alibi_slopes, deterministic=ctx.deterministic,
            attn_type=ctx.attn_type)
        return dq, dk, dv, None, None, None, None, None, None, None, None, None, None, None


def ring_flash_attn_qkvpacked_func(qkv, dropout_p=0.0, softmax_scale=None, causal=False,
                                   window_size=(-1, -1), softcap=0.0, alibi_slopes=None,
                                   deterministic=False, return_attn_probs=False, group=None,
                                   attn_type: AttnType = AttnType.HIP):
    return RingFlashAttnFunc.apply(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], dropout_p, softmax_scale,
                                   causal, window_size, softcap, alibi_slopes, deterministic,
                                   return_attn_probs, group, attn_type, None)


def ring_flash_attn_kvpacked_func(q, kv, dropout_p=0.0, softmax_scale=None, causal=False,
                                  window_size=(-1, -1), softcap=0.0, alibi_slopes=None,
                                  deterministic=False, return_attn_probs=False, group=None,
                                  attn_type: AttnType = AttnType.HIP):
    return RingFlashAttnFunc.apply(q, kv[:, :, 0], kv[:, :, 1], dropout_p, softmax_scale, causal,
                                   window_size, softcap, alibi_slopes, deterministic,
                                   return_attn_probs, group, attn_type, None)


def ring_flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False,
                         window_size=(-1, -1), softcap=0.0, alibi_slopes=None, deterministic=False,
                         return_attn_probs=False, group=None, attn_type: AttnType = AttnType.HIP,
                         attn_processor=None):
    D = q.shape[-1]
    if kernel_head_dim(D) != D:      # a head dim the kernels do not instantiate (e.g. 96): zero-padded copies
        res = ring_flash_attn_func(*pad_head_dim(q, k, v), dropout_p, D ** -0.5 if softmax_scale is None else softmax_scale, causal,
                                   window_size, softcap, alibi_slopes, deterministic, return_attn_probs, group, attn_type, attn_processor)
        return (res[0][..., :D],) + tuple(res[1:]) if isinstance(res, tuple) else res[..., :D]
    if not needs_grad(q, k, v):      # inference / forward-only benchmarks: no autograd node, no saved tensors (~25 us)
        assert alibi_slopes is None
        _check_hot_path_args(dropout_p, (-1, -1), softcap)
        out, lse = ring_flash_attn_forward(
            group, kernel_operand(q), kernel_operand(k), kernel_operand(v),
            softmax_scale=q.shape[-1] ** (-0.5) if softmax_scale is None else softmax_scale, causal=causal,
            window_size=window_size, attn_type=attn_type, attn_processor=attn_processor)
        return out if not return_attn_probs else (out, lse, None)
    return RingFlashAttnFunc.apply(q, k, v, dropout_p, softmax_scale, causal, window_size, softcap,
                                   alibi_slopes, deterministic, return_attn_probs, group, attn_type,
                                   attn_processor)
