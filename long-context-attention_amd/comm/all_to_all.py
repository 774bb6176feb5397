"""Ulysses head <-> sequence all-to-all: same surface as yunchang/comm/all_to_all.py:15-134.

MI355X-first differences (results identical):
  * one HBM pass per exchange instead of two: the reference copies before AND after
    `all_to_all_single` (all_to_all.py:45-49 and :62-65 / :76-84 and :98-100).  Inside this package the
    receive buffer of the head-scatter exchange is used as a strided (B,S,H/P,D) VIEW of its natural
    (S,B,H/P,D) layout -- the attention kernels take strides -- and the send buffer of the
    sequence-scatter exchange is that same layout, which the kernels write directly (the PUBLIC
    functions and autograd Functions return contiguous tensors like the reference's unless asked not to);
  * the remaining pack / unpack is one `usp_copy_rows` launch (16-byte lanes, rows of H/P*D);
  * P == 1 moves no bytes at all (the reference still makes two copies).
The collective itself is `torch.distributed.all_to_all_single` == RCCL over xGMI on ROCm.
"""
from typing import Any, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

from ..kernels.attention import get_block_backend
from . import relay_exchange


def seq_major_empty(B, S, H, D, dtype, device):
    """A (B,S,H,D) view over an (S,B,H,D)-contiguous buffer: the layout both all-to-all directions
    exchange without a copy."""
    return torch.empty((S, B, H, D), dtype=dtype, device=device).transpose(0, 1)


def is_seq_major(x: Tensor) -> bool:
    B, S, H, D = x.shape
    return x.transpose(0, 1).is_contiguous()


def _copy_rows(dst: Tensor, src: Tensor, row_elems: int, sizes, dst_strides, src_strides):
    """dst/src: base tensors; strides in elements.  Device tensors -> usp_copy_rows; host tensors
    (gloo orchestration tests) -> as_strided copy."""
    es = src.element_size()
    if src.is_cuda:
        get_block_backend().copy_rows(dst, src, row_elems * es, list(sizes),
                                      [s * es for s in dst_strides], [s * es for s in src_strides])
    else:
        # host tensors (gloo orchestration tests): hold the copy to what usp_copy_rows takes (include/usp_hip.h: row
        # bytes, every stride and both pointers multiples of 16), so a layout the device path would refuse fails here too
        assert (row_elems * es) % 16 == 0 and all((st * es) % 16 == 0 for st in list(dst_strides) + list(src_strides)) \
            and (dst.storage_offset() * es) % 16 == 0 and (src.storage_offset() * es) % 16 == 0, \
            f"usp_copy_rows needs 16-byte rows / strides / pointers: row {row_elems * es} B, strides {dst_strides} {src_strides}"
        d = torch.as_strided(dst, list(sizes) + [row_elems], list(dst_strides) + [1],
                             dst.storage_offset())
        s = torch.as_strided(src, list(sizes) + [row_elems], list(src_strides) + [1],
                             src.storage_offset())
        d.copy_(s)


def _exchange(send: Tensor, group, use_sync: bool) -> Tensor:
    if relay_exchange.applicable(send, group):       # a pair's exchange striped over the idle mesh links (opt-in)
        recv = relay_exchange.exchange_relayed(send, group)
        if use_sync and send.is_cuda:
            torch.cuda.synchronize()
        return recv
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    if use_sync and send.is_cuda:
        torch.cuda.synchronize()
    return recv


def pack_heads(x: Tensor, P: int) -> Tensor:
    """(B, S/P, H, D) -> send buffer (P, S/P, B, H/P, D): chunk p goes to ulysses rank p."""
    B, Sl, H, D = x.shape
    assert H % P == 0, f"head count {H} not divisible by ulysses degree {P}"
    hp = H // P
    if x.stride(3) != 1 or x.stride(2) != D:
        x = x.contiguous()
    send = torch.empty((P, Sl, B, hp, D), dtype=x.dtype, device=x.device)
    _copy_rows(send, x, hp * D, (P, Sl, B), (Sl * B * hp * D, B * hp * D, hp * D),
               (hp * D, x.stride(1), x.stride(0)))
    return send


def pack_head_group(x5: Tensor, send: Tensor = None, h0: int = 0) -> Tensor:
    """x5: a (B, S/P, P, h, D) VIEW (any strides on the first three dims, (h, D) contiguous) selecting
    h heads per destination rank -> heads [h0, h0+h) of the send buffer (P, S/P, B, Ht, D) (allocated with
    Ht = h when not given).  Used by the head-group pipeline; q, k and v of a group share ONE send buffer
    (Ht = hq + 2 hkv per rank and group), i.e. one collective instead of three."""
    B, Sl, P, h, D = x5.shape
    assert x5.stride(4) == 1 and x5.stride(3) == D
    if send is None:
        send = torch.empty((P, Sl, B, h, D), dtype=x5.dtype, device=x5.device)
    Ht = send.shape[3]
    assert send.shape == (P, Sl, B, Ht, D) and send.is_contiguous() and h0 + h <= Ht
    base = x5.as_strided((1,), (1,), x5.storage_offset())          # element pointer of the view
    dst = send.as_strided((1,), (1,), send.storage_offset() + h0 * D)
    _copy_rows(dst, base, h * D, (P, Sl, B), (Sl * B * Ht * D, B * Ht * D, Ht * D),
               (x5.stride(2), x5.stride(1), x5.stride(0)))
    return send


def unpack_head_group(recv: Tensor, dst5: Tensor, h0: int = 0) -> None:
    """heads [h0, h0+h) of the receive buffer (P, S/P, B, Ht, D) [chunk p = head group of rank p] -> dst5, a
    (B, S/P, P, h, D) VIEW into the full (B, S/P, H, D) result."""
    P, Sl, B, Ht, D = recv.shape
    h = dst5.shape[3]
    assert dst5.stride(4) == 1 and dst5.stride(3) == D and h0 + h <= Ht
    base = dst5.as_strided((1,), (1,), dst5.storage_offset())
    src = recv.as_strided((1,), (1,), recv.storage_offset() + h0 * D)
    _copy_rows(base, src, h * D, (P, Sl, B), (dst5.stride(2), dst5.stride(1), dst5.stride(0)),
               (Sl * B * Ht * D, B * Ht * D, Ht * D))


def pack_seq_into(send: Tensor, h0: int, x: Tensor) -> None:
    """(B, S, h, D) -> heads [h0, h0+h) of the sequence-scatter send buffer (P, S/P, B, Ht, D)."""
    P, Sl, B, Ht, D = send.shape
    h = x.shape[2]
    assert x.shape == (B, P * Sl, h, D) and h0 + h <= Ht and send.is_contiguous()
    if x.stride(3) != 1 or x.stride(2) != D:
        x = x.contiguous()
    dst = send.as_strided((1,), (1,), send.storage_offset() + h0 * D)
    base = x.as_strided((1,), (1,), x.storage_offset())
    _copy_rows(dst, base, h * D, (P * Sl, B), (B * Ht * D, Ht * D), (x.stride(1), x.stride(0)))


def pack_seq_rows(x: Tensor, P: int, lo: int, hi: int) -> Tensor:
    """Rows [lo, hi) of EVERY destination's chunk of (B, S, h, D) -> a send buffer (P, hi - lo, B, h, D) of its own: one piece
    of a row-chunked sequence-scatter exchange (hybrid/async_attn_layer.py: tails).  Destination p owns rows [p S/P, (p+1) S/P)."""
    B, S, h, D = x.shape
    assert S % P == 0 and 0 <= lo < hi <= S // P
    Sl, n = S // P, hi - lo
    if x.stride(3) != 1 or x.stride(2) != D:
        x = x.contiguous()
    send = torch.empty((P, n, B, h, D), dtype=x.dtype, device=x.device)
    base = x.as_strided((1,), (1,), x.storage_offset() + lo * x.stride(1))
    _copy_rows(send, base, h * D, (P, n, B), (n * B * h * D, B * h * D, h * D), (Sl * x.stride(1), x.stride(1), x.stride(0)))
    return send


def view_seq(recv: Tensor) -> Tensor:
    """receive buffer (P, S/P, B, H/P, D) -> (B, S, H/P, D) strided view (no copy)."""
    P, Sl, B, hp, D = recv.shape
    return recv.view(P * Sl, B, hp, D).transpose(0, 1)


def pack_seq(x: Tensor, P: int) -> Tensor:
    """(B, S, H/P, D) -> send buffer (P, S/P, B, H/P, D); free when x is already seq-major."""
    B, S, hp, D = x.shape
    assert S % P == 0, f"sequence {S} not divisible by ulysses degree {P}"
    if is_seq_major(x):
        send = x.transpose(0, 1)                       # already (S,B,hp,D) contiguous: no copy
    else:
        if x.stride(3) != 1 or x.stride(2) != D:
            x = x.contiguous()
        send = torch.empty((S, B, hp, D), dtype=x.dtype, device=x.device)
        _copy_rows(send, x, hp * D, (S, B), (B * hp * D, hp * D), (x.stride(1), x.stride(0)))
    return send.view(P, S // P, B, hp, D)


def unpack_heads(recv: Tensor) -> Tensor:
    """receive buffer (P, S/P, B, H/P, D) [chunk p = head group p] -> (B, S/P, H, D) contiguous."""
    P, Sl, B, hp, D = recv.shape
    H = hp * P
    out = torch.empty((B, Sl, H, D), dtype=recv.dtype, device=recv.device)
    _copy_rows(out, recv, hp * D, (P, Sl, B), (hp * D, H * D, Sl * H * D),
               (Sl * B * hp * D, B * hp * D, hp * D))
    return out


def heads_to_seq(x: Tensor, group, use_sync: bool = False, contiguous: bool = False) -> Tensor:
    """scatter heads / gather sequence: (B, S/P, H, D) -> (B, S, H/P, D)  (all_to_all.py:36-67)."""
    P = dist.get_world_size(group)
    if P == 1:
        return x
    out = view_seq(_exchange(pack_heads(x, P), group, use_sync))
    return out.contiguous() if contiguous else out


def seq_to_heads(x: Tensor, group, use_sync: bool = False) -> Tensor:
    """scatter sequence / gather heads: (B, S, H/P, D) -> (B, S/P, H, D)  (all_to_all.py:69-102)."""
    P = dist.get_world_size(group)
    if P == 1:
        return x
    return unpack_heads(_exchange(pack_seq(x, P), group, use_sync))


def all_to_all_4D(input: torch.Tensor, scatter_idx: int = 2, gather_idx: int = 1, group=None,
                  use_sync: bool = False) -> torch.Tensor:
    """Public function with the reference's semantics (contiguous result)."""
    assert input.dim() == 4, f"input must be 4D tensor, got {input.dim()} and shape {input.shape}"
    if scatter_idx == 2 and gather_idx == 1:
        return heads_to_seq(input, group, use_sync, contiguous=True)
    if scatter_idx == 1 and gather_idx == 2:
        return seq_to_heads(input, group, use_sync)
    raise RuntimeError("scatter_idx must be 1 or 2 and gather_idx must be 1 or 2")


class SeqAllToAll4D(torch.autograd.Function):
    """all_to_all.py:105-134: forward = the exchange, backward = the inverse exchange.  Like the reference the
    result is CONTIGUOUS.  The layers of this package pass `contiguous=False` (a sixth positional argument the
    reference does not have): the head-scatter result is then the strided (B,S,H/P,D) VIEW of the receive buffer,
    which the attention kernels consume directly (they take strides) -- one HBM pass less per exchange."""

    @staticmethod
    def forward(ctx: Any, group, input: Tensor, scatter_idx: int, gather_idx: int,
                use_sync: bool = False, contiguous: bool = True) -> Tensor:
        ctx.group = group
        ctx.scatter_idx = scatter_idx
        ctx.gather_idx = gather_idx
        ctx.use_sync = use_sync
        ctx.contiguous = contiguous
        if scatter_idx == 2 and gather_idx == 1:
            return heads_to_seq(input, group, use_sync, contiguous=contiguous)
        if scatter_idx == 1 and gather_idx == 2:
            return seq_to_heads(input, group, use_sync)
        raise RuntimeError("scatter_idx must be 1 or 2 and gather_idx must be 1 or 2")

    @staticmethod
    def backward(ctx: Any, *grad_output: Tensor) -> Tuple[None, Tensor, None, None, None, None]:
        return (None,
                SeqAllToAll4D.apply(ctx.group, grad_output[0], ctx.gather_idx, ctx.scatter_idx,
                                    ctx.use_sync, ctx.contiguous),
                None, None, None, None)


# --------------------------------------------------------------------------------------------------
# packed qkv: (bs, seq, 3, heads, dim)   (yunchang/comm/all_to_all.py:137-259)
# --------------------------------------------------------------------------------------------------
def pack_heads_5d(x: Tensor, P: int) -> Tensor:
    """(B, S/P, T, H, D) -> send buffer (P, S/P, B, T, H/P, D)."""
    B, Sl, T, H, D = x.shape
    assert H % P == 0, f"head count {H} not divisible by ulysses degree {P}"
    hp = H // P
    if x.stride(4) != 1 or x.stride(3) != D:
        x = x.contiguous()
    send = torch.empty((P, Sl, B, T, hp, D), dtype=x.dtype, device=x.device)
    _copy_rows(send, x, hp * D, (P, Sl, B, T),
               (Sl * B * T * hp * D, B * T * hp * D, T * hp * D, hp * D),
               (hp * D, x.stride(1), x.stride(0), x.stride(2)))
    return send


def view_seq_5d(recv: Tensor) -> Tensor:
    P, Sl, B, T, hp, D = recv.shape
    return recv.view(P * Sl, B, T, hp, D).transpose(0, 1)


def pack_seq_5d(x: Tensor, P: int) -> Tensor:
    B, S, T, hp, D = x.shape
    assert S % P == 0, f"sequence {S} not divisible by ulysses degree {P}"
    if x.transpose(0, 1).is_contiguous():
        send = x.transpose(0, 1)
    else:
        if not x[0, 0].is_contiguous():
            x = x.contiguous()
        send = torch.empty((S, B, T, hp, D), dtype=x.dtype, device=x.device)
        _copy_rows(send, x, T * hp * D, (S, B), (B * T * hp * D, T * hp * D), (x.stride(1), x.stride(0)))
    return send.view(P, S // P, B, T, hp, D)


def unpack_heads_5d(recv: Tensor) -> Tensor:
    P, Sl, B, T, hp, D = recv.shape
    H = hp * P
    out = torch.empty((B, Sl, T, H, D), dtype=recv.dtype, device=recv.device)
    _copy_rows(out, recv, hp * D, (P, Sl, B, T), (hp * D, T * H * D, Sl * T * H * D, H * D),
               (Sl * B * T * hp * D, B * T * hp * D, T * hp * D, hp * D))
    return out


def heads_to_seq_5d(x: Tensor, group, use_sync: bool = False, contiguous: bool = False) -> Tensor:
    """scatter heads / gather sequence: (B, S/P, T, H, D) -> (B, S, T, H/P, D) -- ONE exchange for
    q, k and v (all_to_all.py:160-192)."""
    P = dist.get_world_size(group)
    if P == 1:
        return x
    out = view_seq_5d(_exchange(pack_heads_5d(x, P), group, use_sync))
    return out.contiguous() if contiguous else out


def seq_to_heads_5d(x: Tensor, group, use_sync: bool = False) -> Tensor:
    """scatter sequence / gather heads: (B, S, T, H/P, D) -> (B, S/P, T, H, D) (all_to_all.py:193-233)."""
    P = dist.get_world_size(group)
    if P == 1:
        return x
    return unpack_heads_5d(_exchange(pack_seq_5d(x, P), group, use_sync))


def all_to_all_5D(input: torch.Tensor, scatter_idx: int = 3, gather_idx: int = 1, group=None,
                  use_sync: bool = False) -> torch.Tensor:
    assert input.dim() == 5, f"input must be 5D tensor, got {input.dim()} and shape {input.shape}"
    if scatter_idx == 3 and gather_idx == 1:
        assert input.shape[2] == 3
        return heads_to_seq_5d(input, group, use_sync, contiguous=True)
    if scatter_idx == 1 and gather_idx == 3:
        return seq_to_heads_5d(input, group, use_sync)
    raise RuntimeError("scatter_idx must be 1 or 3 and gather_idx must be 1 or 3")


class SeqAllToAll5D(torch.autograd.Function):
    """all_to_all.py:236-259; `contiguous` as in SeqAllToAll4D."""

    @staticmethod
    def forward(ctx: Any, group, input: Tensor, scatter_idx: int = 3, gather_idx: int = 1,
                use_sync: bool = False, contiguous: bool = True) -> Tensor:
        ctx.group = group
        ctx.scatter_idx = scatter_idx
        ctx.gather_idx = gather_idx
        ctx.use_sync = use_sync
        ctx.contiguous = contiguous
        if scatter_idx == 3 and gather_idx == 1:
            return heads_to_seq_5d(input, group, use_sync, contiguous=contiguous)
        if scatter_idx == 1 and gather_idx == 3:
            return seq_to_heads_5d(input, group, use_sync)
        raise RuntimeError("scatter_idx must be 1 or 3 and gather_idx must be 1 or 3")

    @staticmethod
    def backward(ctx: Any, *grad_output: Tensor) -> Tuple[None, Tensor, None, None, None, None]:
        return (None,
                SeqAllToAll5D.apply(ctx.group, grad_output[0], ctx.gather_idx, ctx.scatter_idx,
                                    ctx.use_sync, ctx.contiguous),
                None, None, None, None)
