"""AsyncLongContextAttention: same surface as yunchang/hybrid/async_attn_layer.py:13-202
 -- the Ulysses all-to-all hidden behind the ring attention by pipelining over
head groups.

The heads each rank owns after the exchange are split into groups (one KV head + its query heads per
group); the exchanges of all groups are queued on a side HIP stream up-front, group i's ring attention
starts as soon as ITS exchange has landed, and its output exchange runs behind group i+1's attention.
Only the first input exchange and the last output exchange stay exposed.

Beyond the reference (which is forward-only `:199-202`, needs Hkv == Hq `:78` and puts one head per rank
in a group): GQA, a backward pass (same pipeline, mirrored), and results that are bit-identical in head
placement to LongContextAttention (group i of rank p = kv head p*Hkv/P + i).
"""
import os
from typing import Any

import torch
import torch.distributed as dist
from torch import Tensor

from ..comm import all_to_all as A
from ..comm.link import link_bytes_per_s
from ..kernels.attention import kernel_head_dim, pad_head_dim
from ..globals import PROCESS_GROUP
from ..kernels import AttnType
from ..ring.ring_flash_attn import ring_flash_attn_backward, ring_flash_attn_forward
from ..ring.stripe_flash_attn import stripe_flash_attn_backward, stripe_flash_attn_forward
from ..ring.utils import _side_stream
from ..ring.zigzag_ring_flash_attn import (_check_hot_path_args, zigzag_forward_phases, zigzag_ring_flash_attn_backward,
                                           zigzag_ring_flash_attn_forward)

_RING_FWD_BWD = {
    "basic": (ring_flash_attn_forward, ring_flash_attn_backward),
    "zigzag": (zigzag_ring_flash_attn_forward, zigzag_ring_flash_attn_backward),
    "strip": (stripe_flash_attn_forward, stripe_flash_attn_backward),
}


class _Lane:
    """The side stream the Ulysses exchanges run on (no-op on host tensors: gloo tests).  Its own lane
    (`_side_stream(device, "ulysses")`): a ring hop queued by KVRelay never waits behind the exchanges of
    later head groups."""

    def __init__(self, ref: Tensor):
        self.cuda = ref.is_cuda
        if self.cuda:
            self.main = torch.cuda.current_stream()
            self.side = _side_stream(ref.device, "ring" if safe_comm() else "ulysses")

    def exchange(self, send: Tensor, group, before=None) -> tuple:
        """Queue all_to_all_single(send) behind everything currently on the main stream; returns
        (recv, event).  `send` must stay referenced until `wait` (the caller keeps it).  `before(side_stream)`: work
        to queue on the lane in front of the collective (it finishes filling `send` from data the MAIN stream does
        not have to wait for: the ring backward's last dK/dV hop)."""
        if not self.cuda:
            if before is not None:
                before(None)
            return A._exchange(send, group, False), None
        ready = torch.cuda.Event()
        ready.record(self.main)
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            if before is not None:
                before(self.side)
            recv = A._exchange(send, group, False)       # (module attribute: bench.py's overlap probe swaps it)
            done = torch.cuda.Event()
            done.record(self.side)
        send.record_stream(self.side)
        recv.record_stream(self.main)
        return recv, done

    def wait(self, event):
        if event is not None:
            torch.cuda.current_stream().wait_event(event)

    def finish(self):
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.side)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.finish()
        return False


_COMM_OVERRIDE = {}      # bench.py / tests: {"safe": bool, "pipeline": "0" | "1" | "auto"} instead of the environment


def safe_comm() -> bool:
    """USP_SAFE_COMM=1: never more than ONE RCCL communicator in flight -- beside a ring (degree > 1) the Ulysses
    exchange is not pipelined (one packed exchange in front of the ring attention, one behind it) and the exchange lane
    shares the ring's side stream, so every collective of a rank is issued in ONE program order on ONE stream.  The
    escape hatch for a first contact with a multi-GPU box: if two communicators with kernels in flight ever stall each
    other (ranks whose kernels reach the device in different orders), this flag costs overlap, not the run."""
    if "safe" in _COMM_OVERRIDE:
        return bool(_COMM_OVERRIDE["safe"])
    return os.environ.get("USP_SAFE_COMM", "0") == "1"


def pipeline_mode(ring_degree: int) -> bool:
    """Is the Ulysses exchange pipelined over head groups (USP_PIPELINE_ULYSSES)?
        0     never: one packed exchange in front of the attention, one behind it;
        1     always, also beside a ring: two communicators (ulysses all-to-all, ring p2p) have kernels in flight at the
              same time, each on its own side stream, every rank issuing them in the same program order;
        auto  (default) everywhere, unless USP_SAFE_COMM=1 (then only at ring degree 1, where there is ONE communicator);
              see `_PIPELINE_BESIDE_RING_DEFAULT`."""
    mode = _COMM_OVERRIDE.get("pipeline", os.environ.get("USP_PIPELINE_ULYSSES", "auto"))
    if mode == "0":
        return False
    if ring_degree > 1 and safe_comm():
        return False
    if mode == "1" or ring_degree == 1:
        return True
    return _PIPELINE_BESIDE_RING_DEFAULT


# Round 5 built the self-chunk start opt-in; round 6 made it the default once it was green through the gloo grids (2 x 1, 2 x 2,
# 2 x 4), two processes on one GPU and both bench workloads at full size on the RCCL virtual grid -- it costs <= 0.15 ms of
# kernel time per iteration at the 8-GPU grid and hides the first exchange of each pass (tools/link_model.py).  Like every
# multi-GPU default of this package it has never met two devices: USP_SELF_CHUNK=0 (or USP_SAFE_COMM=1) is the way back.
_SELF_CHUNK_DEFAULT = "1"


def self_chunk_mode(P: int, ring: int, causal, impl: str, rows_local: int) -> bool:
    """Does a head group start on the rows this rank already holds (USP_SELF_CHUNK; round 5: opt-in, round 6: the default)?

    At ulysses degree 2 half of every exchanged tensor is the self chunk: a rank's own rows of the heads it will own never
    cross a link.  Causal attention over those rows alone is a complete sub-block of the group's work -- rank 0 (rows [0, c)):
    q[0:c] x k[0:c] causal = all of those rows; rank 1 (rows [c, 2c)): the diagonal block q[c:2c] x k[c:2c] -- so the first
    group's kernels start AT ONCE, on views of the send buffer's self chunk, and only the rest of the group waits for the
    exchange (forward: 1/4 resp. 1/4 of the group's work in front of the wait; backward, where K and V are already there and
    only dO travels: 1/4 resp. 3/4).  That hides the one exchange of each pass nothing else can hide -- the first.
    At ring degree 1 (the 2-GPU grid) the ring function is ONE causal block, split here into two or three launches joined
    by the kernel's fused LSE merge / fp32 accumulation (_split_first_forward / _split_first_backward).  Beside a zigzag
    ring (the 8-GPU grid: ulysses 2 x ring 4) the same split applies to STEP 0 of the ring schedule -- the exchange
    delivers exactly one zigzag chunk, rank u = 0 owns the front chunk and u = 1 the back one -- and the ring's K/V
    transfers, which read the exchanged tensors, are posted behind the wait (ring/zigzag_ring_flash_attn.py: `first`).
    Results equal the unsplit launch up to fp32 summation order (the merge is the ring's own)."""
    mode = _COMM_OVERRIDE.get("self_chunk", os.environ.get("USP_SELF_CHUNK", _SELF_CHUNK_DEFAULT))
    if str(mode) not in ("1", "True", "all"):
        return False
    if not (P == 2 and bool(causal) and rows_local >= 1):
        return False
    if ring > 1 and safe_comm():      # (the split backward posts the ring's K/V transfers before it waits for the dO exchange:
        return False                  #  two communicators in flight -- not in the mode whose contract is "one at a time")
    return impl in ("basic", "zigzag") if ring == 1 else impl == "zigzag"


def tails_mode(P: int, ring: int, causal, impl: str, rows_local: int, pipelined: bool, chunk_items: int = None) -> int:
    """Row pieces of the LAST head group's output exchange (0: one exchange behind the group's last kernel, rounds 1-5).

    The head-group pipeline hides every exchange but the first input and the last output of a pass.  The last output waits
    for the last launch of the last group's ring schedule -- and that launch finalises the very rows that travel (a zigzag
    step s > r updates the back chunk; at ulysses degree 2 the back chunk is what ulysses rank 0 sends).  With tails that
    launch runs in n row pieces (ring/zigzag_ring_flash_attn.py:_final_rows; every piece cut along K so that it still fills
    the part) and piece j's rows leave in an exchange of their own while piece j + 1 computes: 1 / n of the exchange stays
    exposed.  The BACKWARD's counterpart needs no pieces: the last ring step issues its dQ launch first and dq -- 4/5 of the
    gradient exchange's bytes at G = 8 -- travels beside the step's dK/dV launch (`dq_first`); what stays exposed is the last
    dK/dV hop and the small dk | dv exchange.  Ulysses degree 2 beside a zigzag ring (the 8-GPU grid), pipelined mode only (in
    the safe mode a second communicator must not start inside a ring pass), and at ring degree 1 (the 2-GPU grid), where the
    last group's one causal block is issued in the layer as 2 n row-range launches (_tail_last_forward) and its backward dQ
    first (_tail_last_backward); USP_TAILS=0 | n overrides (default 4)."""
    mode = _COMM_OVERRIDE.get("tails", os.environ.get("USP_TAILS", "4"))
    n = int(mode)
    ok = P == 2 and bool(causal) and pipelined and (impl == "zigzag" if ring > 1 else impl in ("basic", "zigzag"))
    if n <= 0 or not ok or (ring > 1 and safe_comm()):
        return 0
    if "tails" in _COMM_OVERRIDE:                  # tests: any piece count the rows allow
        return max(1, min(n, rows_local))
    n = max(1, min(n, rows_local // 64))
    # A piece must still fill the part: `chunk_items` = 256-row work items of ONE c-row chunk of the last group (B x its query
    # heads x c / 256).  Beside a ring a piece is cut along K up to 8 times (tail_k_splits), at ring degree 1 it is a plain causal
    # launch over both chunks: pieces of fewer than one item per CU (after the cuts) are not made -- BASELINE's configs[2]
    # (2 GPUs, S16384 H16, two-head groups of 64 items) keeps its one launch per group, the metric's 2-GPU grid (1024 items per
    # chunk and group) gets 4 pieces, the 8-GPU grid (256 items per chunk, x 4 cuts per piece) gets 4.
    if chunk_items is not None:
        from ..comm.link import device_cus
        per_cu = device_cus() // (8 if ring > 1 else 2)
        n = min(n, chunk_items // max(1, per_cu))
    return n if n > 1 else 0


def self_chunk_all_groups() -> bool:
    """USP_SELF_CHUNK=all: beside a ring EVERY head group's owned chunk is launched in front of the first wait, not only the
    first group's.  Measured on one rank of the 8-GPU grid (profiles/r06_rank_emulation.txt): each group started that way costs
    ~0.16 ms of kernel time (quarter-size causal launches cannot balance) and the second group's start hides ~0.13 ms more of the
    0.33 ms first exchange at 64 GB/s -- a wash in time, +0.02 in the overlap figure; off by default."""
    return str(_COMM_OVERRIDE.get("self_chunk", os.environ.get("USP_SELF_CHUNK", _SELF_CHUNK_DEFAULT))) == "all"


def _self_views(send, u, splits):
    """The self chunk of a packed send buffer (P, S/P, B, Ht, D) as (B, S/P, h_j, D) strided views, heads cut at `splits`."""
    full = send[u].transpose(0, 1)
    out, h0 = [], 0
    for h in splits:
        out.append(full[:, :, h0:h0 + h])
        h0 += h
    return out


def _split_first_forward(be, u, selfs, full, wait, scale):
    """The first head group's causal block at ring degree 1, ulysses degree 2, started on the self chunk (self_chunk_mode).
    `selfs` = (q, k, v) of this rank's own rows (views of the send buffer), `full` = (q, k, v) over all 2c rows (views of the
    receive buffer, valid behind `wait()`).  Returns (out, lse) as the ring forward would."""
    qs, ks, vs = selfs
    q, k, v = full
    B, S, hq, D = q.shape
    c = S // 2
    out = torch.empty((B, S, hq, D), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, hq, S), dtype=torch.float32, device=q.device)
    if u == 0:       # own rows [0, c): they see own keys only -- complete and final at once
        be.fwd(qs, ks, vs, scale, True, lse[:, :, :c], out[:, :c])
        wait()
        be.fwd(q[:, c:], k, v, scale, True, lse[:, :, c:], out[:, c:])          # c rows x 2c keys, bottom-right causal
    else:            # own rows [c, 2c): the diagonal block now, the peer's keys merged in behind the exchange
        acc = torch.empty((B, c, hq, D), dtype=torch.float32, device=q.device)
        be.fwd(qs, ks, vs, scale, True, lse[:, :, c:], None, acc, False, 0, 0)
        wait()
        be.fwd(q[:, c:], k[:, :c], v[:, :c], scale, False, lse[:, :, c:], out[:, c:], acc, True, 0, c)
        be.fwd(q[:, :c], k[:, :c], v[:, :c], scale, True, lse[:, :, :c], out[:, :c])
    return out, lse


def _split_first_backward(be, u, do_self, do_full, wait, q, k, v, o, lse, scale):
    """The backward of the same block: K, V, out and the LSE are there (saved), only dO travels -- the rows this rank owns
    start at once, the peer's rows follow behind the exchange; dK / dV accumulate in fp32 across the two launches and are
    rounded by the last one that touches a row.  Returns (dq, dk, dv) in q.dtype."""
    B, S, hq, D = q.shape
    kvh = k.shape[2]
    c = S // 2
    dev = q.device
    delta = torch.empty((B, hq, S), dtype=torch.float32, device=dev)
    dq = torch.empty((B, S, hq, D), dtype=q.dtype, device=dev)
    dk = torch.empty((B, S, kvh, D), dtype=k.dtype, device=dev)
    dv = torch.empty_like(dk)
    dk32 = torch.empty((B, S, kvh, D), dtype=torch.float32, device=dev)
    dv32 = torch.empty_like(dk32)
    if u == 0:       # rows [0, c) x keys [0, c) first; then rows [c, 2c) x all keys on top
        be.delta(do_self, o[:, :c], delta[:, :, :c])
        dk32[:, c:].zero_()
        dv32[:, c:].zero_()
        be.bwd(do_self, q[:, :c], k[:, :c], v[:, :c], lse[:, :, :c], delta[:, :, :c], None, dk32[:, :c], dv32[:, :c], scale, True,
               dq16=dq[:, :c])
        wait()
        be.delta(do_full[:, c:], o[:, c:], delta[:, :, c:])
        be.bwd(do_full[:, c:], q[:, c:], k, v, lse[:, :, c:], delta[:, :, c:], None, dk32, dv32, scale, True,
               accum_dk=True, accum_dv=True, dq16=dq[:, c:], dk16=dk, dv16=dv)
    else:            # rows [c, 2c) x all keys first (3/4 of the block); then rows [0, c) x keys [0, c) on top
        be.delta(do_self, o[:, c:], delta[:, :, c:])
        be.bwd(do_self, q[:, c:], k, v, lse[:, :, c:], delta[:, :, c:], None, dk32, dv32, scale, True, dq16=dq[:, c:])
        wait()
        be.delta(do_full[:, :c], o[:, :c], delta[:, :, :c])
        be.bwd(do_full[:, :c], q[:, :c], k[:, :c], v[:, :c], lse[:, :, :c], delta[:, :, :c], None, dk32[:, :c], dv32[:, :c], scale,
               True, accum_dk=True, accum_dv=True, dq16=dq[:, :c], dk16=dk[:, :c], dv16=dv[:, :c])
        be.cast(dk[:, c:], dk32[:, c:])              # keys [c, 2c) got gradients from the first launch only
        be.cast(dv[:, c:], dv32[:, c:])
    return dq, dk, dv


def _tail_last_forward(be, q, k, v, scale, n, emit):
    """The LAST head group's causal block at ring degree 1 (ulysses degree 2) in row pieces: piece j of the front chunk, piece j
    of the back chunk (each a bottom-right-aligned causal launch over the keys its rows see), then `emit(j, out)` -- the piece's
    output exchange runs beside the next piece's launches.  Same rows x keys as one launch; returns (out, lse)."""
    B, S, hq, D = q.shape
    c = S // 2
    out = torch.empty((B, S, hq, D), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, hq, S), dtype=torch.float32, device=q.device)
    for j in range(n):
        for ch in (0, 1):
            a, b = ch * c + j * c // n, ch * c + (j + 1) * c // n
            if b > a:
                be.fwd(q[:, a:b], k[:, :b], v[:, :b], scale, True, lse[:, :, a:b], out[:, a:b])
        emit(j, out)
    return out, lse


def _tail_last_backward(be, dout, q, k, v, o, lse, scale, dq_first):
    """... and its backward: delta, the dQ launch (rounded in its epilogue), `dq_first(dq)` -- dq travels beside the dK/dV launch
    --, the dK/dV launch.  Returns (dq, dk, dv) in q.dtype."""
    B, S, hq, D = q.shape
    delta = torch.empty((B, hq, S), dtype=torch.float32, device=q.device)
    dq = torch.empty((B, S, hq, D), dtype=q.dtype, device=q.device)
    dk = torch.empty((B, S, k.shape[2], D), dtype=k.dtype, device=q.device)
    dv = torch.empty_like(dk)
    be.delta(dout, o, delta)
    be.bwd(dout, q, k, v, lse, delta, None, None, None, scale, True, dq16=dq, only="dq")
    dq_first(dq)
    be.bwd(dout, q, k, v, lse, delta, None, None, None, scale, True, dk16=dk, dv16=dv, only="dkdv")
    return dq, dk, dv


# The default beside a ring.  Round 2 turned it on without any run through RCCL; round 3 first
# built tests/test_gpu_rccl_order.py::test_pipelined_exchange_beside_a_ring_through_rccl -- BASELINE's 8-GPU grid
# (ulysses 2 x ring 4, GQA, zigzag, forward + backward) and the 2 x 2 grids as virtual ranks whose every exchange and
# every ring transfer is a real RCCL call, stream-ordered only, bit-identical over iterations and equal to the
# reference's run -- and turned it on when that was green on an MI355X (gpurun_out/r03/5_rccl_order.log).  What the
# test cannot show is two communicators across eight DEVICES: USP_SAFE_COMM=1 is the escape hatch, and bench.py
# --gpus 8 measures the safe mode first and the overlapped mode under a deadline.
_PIPELINE_BESIDE_RING_DEFAULT = True

_MAX_GROUPS = 4     # deeper pipelines only shrink the per-group kernels (fewer workgroups per launch)


# 256-row work items a head group's attention launch must still have, in units of the device's CU count (comm/link.py:
# device_cus; 256 on an MI355X).  Measured (kbench, C3 shape = one causal launch per group, interleavable): 8 heads x 64
# blocks = 512 items (2 per CU) run at 1040-1145 TFLOP/s, 4 heads = 256 items (1 per CU) at 814 (905 since the tiles of a head
# are dealt evenly to its XCDs), 2 heads at 536-565 -- with one item per CU nothing balances the causal triangle.  So TWO
# items per CU are asked for ...
_FILL_PER_CU = 2.0
# ... unless the exchange is long against the attention it can hide behind: then the pipeline wins even with starved
# launches.  C3 (2 GPUs, forward only, MHA): 50 MB in + 17 MB out per rank over ONE link = 1.05 ms against 0.5 ms of
# attention -- sequential 0.79 + 0.54 + 0.26 = 1.59 ms, two groups of 256 items (0.36 ms each) 1.28 ms, four groups
# 1.33 ms (the schedule: all input exchanges queued first on the lane, group i's output behind its attention).  The
# one-GPU rank emulation cannot see this (its wire is an HBM copy): its 0.69 vs 0.86 ms is kernel time only.  ONE per CU ...
_FILL_PER_CU_LINK_BOUND = 1.0
# ... and half of that again where the forward kernel cuts few-item launches along K (usp_fwd_args.k_splits, staged behind
# USP_FWD_KSPLIT): a 2-head group of the 2-GPU config (128 items) then runs at 936 instead of 557 TFLOP/s (kbench ksplit),
# and four such groups put the iteration on the wire's floor (1.05 ms against 1.22 with two groups, tools/link_model.py).
# Forward-only calls: the backward kernels have no such cut.
_FILL_PER_CU_LINK_BOUND_KSPLIT = 0.5
_FILL_ITEMS = None            # tests pin an item count here (1: let the pipeline form on tiny problems)
_LINK_BYTES_PER_S = None      # tests pin a rate here; otherwise comm/link.py: measured at set_seq_parallel_pg time, or 64 GB/s
_KERNEL_FLOPS_PER_S = None    # likewise: comm/link.py: measured beside the link (USP_LINK_PROBE=1), or 1.1e15


def fill_items(link_bound=False, k_split=False):
    """Work items a head group's launch must still have (see the three per-CU figures above)."""
    if _FILL_ITEMS is not None:
        return _FILL_ITEMS
    from ..comm.link import device_cus
    per_cu = _FILL_PER_CU if not link_bound else (_FILL_PER_CU_LINK_BOUND_KSPLIT if k_split else _FILL_PER_CU_LINK_BOUND)
    return max(1, int(per_cu * device_cus()))


def _link_bound(Hq, Hkv, P, B, S, D, itemsize, ring, causal):
    """Is the Ulysses exchange of one forward pass at least half as long as the attention it surrounds?  Every rank
    sends 1/P of its local q|k|v to each of its P-1 peers over that peer's own link, and gets 1/P of the output back."""
    rows = B * (S // P)                                              # local rows before the exchange
    t_comm = rows * (2 * Hq + 2 * Hkv) * D * itemsize / P / (_LINK_BYTES_PER_S or link_bytes_per_s())
    flops = 4.0 * B * (Hq // P) * S * (S * ring) * D * (0.5 if causal else 1.0)
    from ..comm.link import kernel_flops_per_s
    return t_comm >= 0.5 * flops / (_KERNEL_FLOPS_PER_S or kernel_flops_per_s())


def _groups(Hq, Hkv, P, B=None, S=None, max_groups=None, link_bound=False, k_split=False):
    """(number of head groups, kv heads per rank per group, query heads per kv head).  A group is a
    set of whole KV heads (with their query heads) of every rank's post-exchange share.  With the problem
    size (B, S = sequence after the exchange) given, the pipeline is kept shallow enough that every group's
    attention launch still has `fill_items` 256-row work items (two per CU; one when the caller found the exchange long
    against the attention, `_link_bound`; half of one when, in addition, the forward kernel will cut such launches along
    K, `k_split`)."""
    assert Hq % P == 0 and Hkv % P == 0, f"heads ({Hq}, {Hkv}) not divisible by ulysses degree {P}"
    per_rank = Hkv // P
    ng = 1
    if P > 1:                       # nothing to hide without an exchange
        cap = _MAX_GROUPS if max_groups is None else min(_MAX_GROUPS, max_groups)
        if B is not None and S is not None:
            fill = fill_items(link_bound, k_split)
            cap = max(1, min(cap, (B * (Hq // P) * ((S + 255) // 256)) // fill))
        for cand in range(min(cap, per_rank), 0, -1):
            if per_rank % cand == 0:
                ng = cand
                break
    return ng, per_rank // ng, Hq // Hkv


def _k_split_groups(ctx, B, S, causal):
    """May the head groups be sized for launches the forward kernel cuts along K?  Only where that cut is on (the staged
    USP_FWD_KSPLIT policy answers for a 2-head launch of this length) and no backward will follow."""
    from .. import _C
    if ctx is not None and any(ctx.needs_input_grad[:3]):
        return False
    return _C.fwd_k_splits(B, S, 2, bool(causal)) > 1


def _qkv_to_seq(lane, q, k, v, P, ng, kvh, g, i, group):
    """ONE exchange for head group i of q, k and v (B, S/P, H{q,kv}, D): the send buffer is
    (P, S/P, B, kvh*g + 2*kvh, D) = [q heads | k heads | v heads] of (destination rank, group i) -- the
    GQA-capable form of the reference's packed exchange (async_attn_layer.py:100-128 stacks q|k|v of one head
    per rank, which needs Hkv == Hq).  Returns ((q, k, v) as (B, S, h, D) strided views of the receive buffer,
    event, the send buffer)."""
    hq = kvh * g
    send = None
    for x, h, h0 in ((q, hq, 0), (k, kvh, hq), (v, kvh, hq + kvh)):
        B, Sl, _, D = x.shape
        if x.stride(3) != 1 or x.stride(2) != D:
            x = x.contiguous()
        if send is None:
            send = torch.empty((P, Sl, B, hq + 2 * kvh, D), dtype=x.dtype, device=x.device)
        A.pack_head_group(x.view(B, Sl, P, ng, h, D)[:, :, :, i], send, h0)    # heads p*(ng*h) + i*h + (0..h)
    recv, ev = lane.exchange(send, group)
    full = A.view_seq(recv)                                        # (B, S, hq + 2 kvh, D)
    return (full[:, :, :hq], full[:, :, hq:hq + kvh], full[:, :, hq + kvh:]), ev, send


def _to_seq(lane, x, P, ng, h, i, group):
    """Exchange head group i of x (B, S/P, H, D): returns ((B, S, h, D) view, event, the send buffer)."""
    B, Sl, H, D = x.shape
    if x.stride(3) != 1 or x.stride(2) != D:
        x = x.contiguous()
    x5 = x.view(B, Sl, P, ng, h, D)[:, :, :, i]                   # heads p*(ng*h) + i*h + (0..h)
    send = A.pack_head_group(x5)
    recv, ev = lane.exchange(send, group)
    return A.view_seq(recv), ev, send


def _to_heads_issue(lane, xs, P, group):
    """Queue ONE exchange of the (B, S, h_j, D) tensors `xs` back to sequence sharding (heads stacked in one
    send buffer); returns (recv (P, S/P, B, sum h_j, D), event)."""
    if len(xs) == 1:
        return lane.exchange(A.pack_seq(xs[0], P), group)
    B, S, _, D = xs[0].shape
    Ht = sum(x.shape[2] for x in xs)
    send = torch.empty((P, S // P, B, Ht, D), dtype=xs[0].dtype, device=xs[0].device)
    h0 = 0
    for x in xs:
        A.pack_seq_into(send, h0, x)
        h0 += x.shape[2]
    return lane.exchange(send, group)


def _grads_to_heads_issue(lane, dq, dk, dv, tail, P, group):
    """ONE exchange dq | dk | dv of a head group back to sequence sharding (dq None: dk | dv alone -- dq went ahead in an
    exchange of its own, `dq_first`), with the ring backward's LAST dK/dV hop
    still in flight (`tail`: the pending RingComm of travel_dkdv's `defer`): dq is packed on the compute stream, which
    then goes on to the next group's kernels; the lane waits for the hop, packs dk and dv behind it and runs the
    collective.  What the compute stream used to wait for (16 MiB of fp32 per KV head over one link, 0.26 ms at 64 GB/s
    per group at BASELINE's 8-GPU config) now runs beside the next group's first ring step."""
    B, S, kvh, D = dk.shape
    hq = 0 if dq is None else dq.shape[2]
    send = torch.empty((P, S // P, B, hq + 2 * kvh, D), dtype=dk.dtype, device=dk.device)
    if dq is not None:
        A.pack_seq_into(send, 0, dq)

    def before(side):
        for comm in tail:
            comm.wait()                                  # stream-wise on a GPU: the LANE waits for the hop
            if side is not None:
                for t in getattr(comm, "keep", ()):      # the hop's send buffers: not to be reused before it is through
                    t.record_stream(side)
        if side is not None:
            dk.record_stream(side)
            dv.record_stream(side)
        A.pack_seq_into(send, hq, dk)
        A.pack_seq_into(send, hq + kvh, dv)
    return lane.exchange(send, group, before)


class _AsyncUSPFunc(torch.autograd.Function):
    """forward/backward of the USP layer with packed, optionally pipelined exchanges.  `ng_cap` = 1 keeps the
    sequential order (one packed exchange in, attention, one exchange out)."""

    @staticmethod
    def forward(ctx, q, k, v, softmax_scale, causal, ulysses_pg, ring_pg, impl, ng_cap=None):
        fwd, _ = _RING_FWD_BWD[impl]
        P = dist.get_world_size(ulysses_pg)
        B, Sl, Hq, D = q.shape
        Hkv = k.shape[2]
        ring = dist.get_world_size(ring_pg) if ring_pg is not None else 1
        ng, kvh, g = _groups(Hq, Hkv, P, B, Sl * P, max_groups=ng_cap,
                             link_bound=P > 1 and _link_bound(Hq, Hkv, P, B, Sl * P, D, q.element_size(), ring, causal),
                             k_split=_k_split_groups(ctx, B, Sl * P, causal))
        if softmax_scale is None:
            softmax_scale = D ** (-0.5)
        overlap = ng > 1                # kernels run beside later groups' exchanges
        split0 = self_chunk_mode(P, ring, causal, impl, Sl)          # groups start on this rank's own rows
        u = dist.get_rank(ulysses_pg) if split0 else 0
        n_tail = tails_mode(P, ring, causal, impl, Sl, ng_cap is None or ng_cap > 1, B * kvh * g * ((Sl + 255) // 256))
        saved, outs = [], []
        with _Lane(q) as lane:
            # every input exchange is queued before any attention runs
            ins = [_qkv_to_seq(lane, q, k, v, P, ng, kvh, g, i, ulysses_pg) for i in range(ng)]
            pieces = []                       # the last group's output exchange in row pieces: (recv, event, lo, hi)

            def tail_of(i):
                if not n_tail or i != ng - 1 or (ring == 1 and split0 and i == 0):     # (ring degree 1, one group: the self-chunk
                    return None                                                      #  split owns that block)

                def emit(j, out_i):           # piece j of both chunks is final on this stream: its exchange starts now
                    lo, hi = j * Sl // n_tail, (j + 1) * Sl // n_tail
                    if hi > lo:
                        pieces.append(lane.exchange(A.pack_seq_rows(out_i, P, lo, hi), ulysses_pg) + (lo, hi))
                return n_tail, emit
            # Beside a ring the first group's owned chunk is launched before the first wait (round 5); USP_SELF_CHUNK=all: every
            # group's (the first exchange of the pass takes longer than one group's owned chunk, and the other groups' owned
            # chunks are the only other work that needs no exchanged byte -- self_chunk_all_groups for what that buys).
            gens = {}
            if split0 and ring > 1:
                for i in range(ng if self_chunk_all_groups() else 1):
                    (qi, ki, vi), ev, send_i = ins[i]
                    own = _self_views(send_i, u, (kvh * g, kvh, kvh))
                    gens[i] = zigzag_forward_phases(ring_pg, qi, ki, vi, softmax_scale, overlap,
                                                    (u, own, lambda ev=ev: lane.wait(ev)), tail_of(i))
                    next(gens[i])             # allocations + the launch on the owned chunk; stops in front of the wait
            for i in range(ng):
                (qi, ki, vi), ev, send_i = ins[i]
                if i in gens:                 # the rest of the ring schedule, behind this group's exchange
                    try:
                        next(gens[i])
                        raise AssertionError("zigzag_forward_phases yields once")
                    except StopIteration as done:
                        oi, lse_i = done.value
                elif split0 and i == 0:       # ring degree 1: the one causal block, split in the layer
                    from ..kernels.attention import get_block_backend
                    own = _self_views(send_i, u, (kvh * g, kvh, kvh))
                    oi, lse_i = _split_first_forward(get_block_backend(beside_transfers=True), u, own, (qi, ki, vi),
                                                     lambda ev=ev: lane.wait(ev), softmax_scale)
                elif ring == 1 and tail_of(i) is not None:       # the 2-GPU grid: the last group's one causal block in row pieces
                    from ..kernels.attention import get_block_backend
                    lane.wait(ev)
                    oi, lse_i = _tail_last_forward(get_block_backend(beside_transfers=True), qi, ki, vi, softmax_scale, *tail_of(i))
                else:
                    lane.wait(ev)
                    t = tail_of(i)
                    kw = {} if t is None else {"tail": t}
                    oi, lse_i = fwd(ring_pg, qi, ki, vi, softmax_scale=softmax_scale, causal=causal, overlap=overlap, **kw)
                saved += [qi, ki, vi, oi, lse_i]
                if tail_of(i) is not None:
                    outs.append(None)         # (travelled in `pieces`)
                else:
                    outs.append(_to_heads_issue(lane, [oi], P, ulysses_pg))
            out = torch.empty((B, Sl, Hq, D), dtype=q.dtype, device=q.device)
            o5 = out.view(B, Sl, P, ng, kvh * g, D)
            for i, pending in enumerate(outs):
                if pending is None:
                    for recv, ev, lo, hi in pieces:
                        lane.wait(ev)
                        A.unpack_head_group(recv, o5[:, lo:hi, :, i])
                    continue
                recv, ev = pending
                lane.wait(ev)
                A.unpack_head_group(recv, o5[:, :, :, i])
        ctx.save_for_backward(*saved)
        ctx.meta = (softmax_scale, causal, ulysses_pg, ring_pg, impl, P, ng, kvh, g, Hq, Hkv)
        ctx.split0 = (split0, u, ring)
        ctx.n_tail = n_tail if tail_of(ng - 1) is not None else 0
        return out

    @staticmethod
    def backward(ctx, dout):
        softmax_scale, causal, ulysses_pg, ring_pg, impl, P, ng, kvh, g, Hq, Hkv = ctx.meta
        _, bwd = _RING_FWD_BWD[impl]
        saved = ctx.saved_tensors
        B, Sl, _, D = dout.shape
        overlap = ng > 1
        with _Lane(dout) as lane:
            douts = [_to_seq(lane, dout, P, ng, kvh * g, i, ulysses_pg) for i in range(ng)]
            pend = []
            split0, u, ring = getattr(ctx, "split0", (False, 0, 1))
            n_tail = getattr(ctx, "n_tail", 0)
            for i in range(ng):
                qi, ki, vi, oi, lse_i = saved[5 * i:5 * i + 5]
                doi, ev, send_i = douts[i]
                kw = {}
                dq_sent = []
                if n_tail and i == ng - 1:     # the last group: dq leaves between the last step's two launches (tails_mode)
                    kw["dq_first"] = lambda dq16: dq_sent.append(_to_heads_issue(lane, [dq16], P, ulysses_pg))
                if split0 and i == 0:
                    from ..kernels.attention import get_block_backend
                    do_own = _self_views(send_i, u, (kvh * g,))[0]
                    tail = []
                    if ring == 1:
                        dqi, dki, dvi = _split_first_backward(get_block_backend(beside_transfers=True), u, do_own, doi,
                                                              lambda ev=ev: lane.wait(ev), qi, ki, vi, oi, lse_i, softmax_scale)
                    else:
                        dqi, dki, dvi = bwd(ring_pg, doi, qi, ki, vi, oi, lse_i, softmax_scale=softmax_scale, causal=causal,
                                            overlap=overlap, tail=tail, first=(u, do_own, lambda ev=ev: lane.wait(ev)), **kw)
                elif ring == 1 and kw:         # the 2-GPU grid's last group: dQ launch, dq exchange, dK/dV launch
                    from ..kernels.attention import get_block_backend
                    lane.wait(ev)
                    tail = []
                    dqi, dki, dvi = _tail_last_backward(get_block_backend(beside_transfers=True), doi, qi, ki, vi, oi, lse_i,
                                                        softmax_scale, kw["dq_first"])
                else:
                    lane.wait(ev)
                    tail = []                  # the ring backward's last dK/dV hop, left pending (ring/utils.py:travel_dkdv)
                    dqi, dki, dvi = bwd(ring_pg, doi, qi, ki, vi, oi, lse_i, softmax_scale=softmax_scale,
                                        causal=causal, overlap=overlap, tail=tail, **kw)
                if dq_sent:                    # two exchanges: dq (posted inside the ring backward), then dk | dv
                    pend.append((dq_sent[0], _grads_to_heads_issue(lane, None, dki, dvi, tail, P, ulysses_pg)))
                else:
                    pend.append(_grads_to_heads_issue(lane, dqi, dki, dvi, tail, P, ulysses_pg))   # ONE exchange: dq | dk | dv
            dq = torch.empty((B, Sl, Hq, D), dtype=dout.dtype, device=dout.device)
            dk = torch.empty((B, Sl, Hkv, D), dtype=dout.dtype, device=dout.device)
            dv = torch.empty_like(dk)
            q5 = dq.view(B, Sl, P, ng, kvh * g, D)
            k5, v5 = dk.view(B, Sl, P, ng, kvh, D), dv.view(B, Sl, P, ng, kvh, D)
            hq = kvh * g
            for i, item in enumerate(pend):
                if isinstance(item[0], tuple):          # (dq exchange, dk | dv exchange)
                    (rq, eq), (recv, ev) = item
                    lane.wait(eq)
                    A.unpack_head_group(rq, q5[:, :, :, i], 0)
                    lane.wait(ev)
                    A.unpack_head_group(recv, k5[:, :, :, i], 0)
                    A.unpack_head_group(recv, v5[:, :, :, i], kvh)
                    continue
                recv, ev = item
                lane.wait(ev)
                A.unpack_head_group(recv, q5[:, :, :, i], 0)
                A.unpack_head_group(recv, k5[:, :, :, i], hq)
                A.unpack_head_group(recv, v5[:, :, :, i], hq + kvh)
        return dq, dk, dv, None, None, None, None, None, None


class AsyncLongContextAttention(torch.nn.Module):
    """Arguments (identical to the reference): scatter_idx, gather_idx, ring_impl_type."""

    def __init__(self, scatter_idx: int = 2, gather_idx: int = 1, ring_impl_type: str = "basic") -> None:
        super(AsyncLongContextAttention, self).__init__()
        self.ring_pg = PROCESS_GROUP.RING_PG
        self.ulysses_pg = PROCESS_GROUP.ULYSSES_PG
        assert (
            self.ulysses_pg is not None or self.ring_pg is not None
        ), f"use set_seq_parallel_pg() first. Now ulysses pg {self.ulysses_pg} and ring pg {self.ring_pg}"
        self.scatter_idx = scatter_idx
        self.gather_idx = gather_idx
        if ring_impl_type not in _RING_FWD_BWD:
            raise KeyError(ring_impl_type)
        self.ring_impl_type = ring_impl_type

    def forward(self, query: Tensor, key: Tensor, value: Tensor, dropout_p=0.0, softmax_scale=None,
                causal=False, window_size=(-1, -1), softcap=0.0, alibi_slopes=None, deterministic=False,
                return_attn_probs=False, *args: Any) -> Tensor:
        """query (bs, seqlen/P, hc, hs); key/value (bs, seqlen/P, hc_kv, hs) -> (bs, seqlen/P, hc, hs)."""
        assert alibi_slopes is None
        _check_hot_path_args(dropout_p, window_size, softcap)
        D = query.shape[-1]
        if kernel_head_dim(D) != D:      # a head dim the kernels do not instantiate: zero-padded copies
            query, key, value = pad_head_dim(query, key, value)
            softmax_scale = D ** -0.5 if softmax_scale is None else softmax_scale
        out = _AsyncUSPFunc.apply(query, key, value, softmax_scale, causal, self.ulysses_pg, self.ring_pg,
                                  self.ring_impl_type)
        return out[..., :D]
