"""Ring plumbing of the USP path: same surface as yunchang/ring/utils.py (RingComm :118-161,
update_out_and_lse :10-51), plus the side-stream relay this package's schedules use.

Transport is torch.distributed point-to-point (`batch_isend_irecv` == grouped RCCL send/recv over
xGMI on ROCm; gloo on CPU for the orchestration tests).
"""
import os
from collections import OrderedDict
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist
import torch.distributed as _torch_dist      # `dist` is what the virtual-rank tests swap; this name never is

from ..kernels.attention import get_block_backend

__all__ = ["update_out_and_lse", "group_info", "RingComm", "KVRelay", "ZigzagKVFetch", "zigzag_fetch_pieces", "zigzag_wave_steps", "clear_slot_caches", "kv_relay_mode", "travel_dkdv", "return_dkdv_direct",
           "dkdv_return_mode", "FULL", "final_grads"]


def update_out_and_lse(out: Optional[torch.Tensor], lse: Optional[torch.Tensor],
                       block_out: torch.Tensor, block_lse: torch.Tensor, slice_=None
                       ) -> Tuple[torch.Tensor, torch.Tensor]:
    """ring/utils.py:30-51 as ONE kernel (usp_lse_merge) instead of ~6 elementwise passes.
    out (B,S,H,D) fp32, lse (B,S,H,1) fp32 [the reference's layout], block_lse (B,H,Sb).
    The schedules of this package do not call this (the merge is fused into the attention
    kernel's epilogue); it is kept for callers of the reference's helper."""
    be = get_block_backend()
    blk_lse = block_lse if block_lse.dtype == torch.float32 else block_lse.float()
    if blk_lse.stride(-1) != 1:
        blk_lse = blk_lse.contiguous()
    if out is None:
        if slice_ is not None:
            raise RuntimeError("first update_out_and_lse should not pass slice_ args")
        B, S, H, D = block_out.shape
        out = torch.empty((B, S, H, D), dtype=torch.float32, device=block_out.device)
        lse_bhs = torch.empty((B, H, S), dtype=torch.float32, device=block_out.device)
        be.merge(out, lse_bhs, block_out, blk_lse, True)
        return out, lse_bhs.transpose(1, 2).unsqueeze(-1)
    lse_bhs = lse.squeeze(-1).transpose(1, 2)          # (B,H,S) view, unit seq stride if we made it
    if lse_bhs.stride(-1) != 1:
        lse_c = lse_bhs.contiguous()
    else:
        lse_c = lse_bhs
    o_view, l_view = out, lse_c
    if slice_ is not None:
        o_view = out[slice_]
        l_view = lse_c[:, :, slice_[1]]
    be.merge(o_view, l_view, block_out, blk_lse, False)
    if lse_c is not lse_bhs:
        lse = lse_c.transpose(1, 2).unsqueeze(-1)
    return out, lse


_GROUP_INFO = {}


def forget_groups():
    """Drop the cached (size, rank) pairs -- and with them the strong references to the ProcessGroup objects they are
    keyed by.  Called by set_seq_parallel_pg: after destroy_process_group + re-init (tests, elastic restarts) the old
    groups and their communicators must be collectable."""
    _GROUP_INFO.clear()


def group_info(dist_mod, process_group):
    """(size, this rank) of a process group.  A group's shape never changes, so torch.distributed is asked once per
    group object (two python-level lookups per launch otherwise, ~10 us of the N = 1 step's 65, tools/host_step_cpu.py).
    `dist_mod` is the CALLER's `dist`: the virtual-rank tests swap that attribute for a stand-in whose rank is per
    thread -- anything but the real module is asked every time; so is the default group (None), which tests re-create."""
    if dist_mod is not _torch_dist or process_group is None:
        return dist_mod.get_world_size(process_group), dist_mod.get_rank(process_group)
    info = _GROUP_INFO.get(process_group)
    if info is None:
        if len(_GROUP_INFO) >= 64:
            _GROUP_INFO.clear()
        info = _GROUP_INFO[process_group] = (_torch_dist.get_world_size(process_group), _torch_dist.get_rank(process_group))
    return info


class RingComm:
    """ring/utils.py:118-161.  send_recv queues an isend to ring rank+1 and an irecv from ring
    rank-1; commit posts them as one batch; wait blocks (stream-wise on a GPU) until done."""

    def __init__(self, process_group: dist.ProcessGroup):
        self._process_group = process_group
        self._ops = []
        self.rank = dist.get_rank(self._process_group)
        self.world_size = dist.get_world_size(self._process_group)
        self._reqs = None
        self.send_rank = (self.rank + 1) % self.world_size
        self.recv_rank = (self.rank - 1) % self.world_size
        if process_group is not None:
            self.send_rank = dist.get_global_rank(self._process_group, self.send_rank)
            self.recv_rank = dist.get_global_rank(self._process_group, self.recv_rank)

    def send_recv(self, to_send: torch.Tensor, recv_tensor: Optional[torch.Tensor] = None) -> torch.Tensor:
        res = torch.empty_like(to_send) if recv_tensor is None else recv_tensor
        self._ops.append(dist.P2POp(dist.isend, to_send, self.send_rank, group=self._process_group))
        self._ops.append(dist.P2POp(dist.irecv, res, self.recv_rank, group=self._process_group))
        return res

    def commit(self):
        if self._reqs is not None:
            raise RuntimeError("commit called twice")
        self._reqs = dist.batch_isend_irecv(self._ops)

    def wait(self):
        if self._reqs is None:
            raise RuntimeError("wait called before commit")
        for req in self._reqs:
            req.wait()
        self._reqs = None
        self._ops = []


_MAX_SLOT_SETS = 8


def _cached_slots(cache: "OrderedDict", key, make):
    """Persistent receive slots, least-recently-used sets dropped beyond _MAX_SLOT_SETS: a training run with a few fixed
    shapes never reallocates, one with ever-changing token counts (packed batches) does not grow without bound.  Dropping
    a set is safe at any time: a live relay holds its own reference, and the memory goes back to the caching allocator
    of the compute stream, on which every kernel that read the slots was queued."""
    slots = cache.get(key)
    if slots is None:
        slots = cache[key] = make()
        while len(cache) > _MAX_SLOT_SETS:
            cache.popitem(last=False)
    else:
        cache.move_to_end(key)
    return slots


class KVRelay:
    """Brings the K and V of every other ring rank to this rank AHEAD of the attention kernels.

    The reference posts the transfer for step s+1 from the compute stream at the top of step s
    (zigzag_ring_flash_attn.py:46-49), so RCCL only starts it once attention(s-1) has finished, and it relays
    hop by hop: the K/V of rank r-s reach rank r over s consecutive hops on ONE link per GPU.  Here everything
    is queued up-front on a side HIP stream (`_side_stream(device, "ring")`), with one receive slot per source
    rank, and the compute stream only waits on the event of the slot it is about to consume:

      * "direct" (default for ring degree > 2): every rank sends its OWN K/V straight to all P-1 peers in ONE
        grouped send/recv.  xGMI is a full mesh of point-to-point links, so the P-1 transfers into a rank
        arrive over P-1 different links in parallel: the same bytes per rank as the relay, but one transfer
        time instead of P-1 serial ones (BASELINE's 4-GPU config moves 64 MiB per hop against 0.25 ms of
        attention per step: a one-link relay is link-bound there).  The transfers of one group run
        concurrently and land together, so all slots share ONE event (ProcessGroupNCCL serialises separate
        batches on its internal stream: per-peer batches would give per-slot events at the price of
        serialising the links again).
      * "chain" (ring degree 2, or USP_KV_RELAY=chain): the reference's hop-by-hop relay, hop s+1 starting the
        moment hop s has landed; one event per slot.

    Slot s holds the K/V of ring rank r-s in both modes.  The receive slots are persistent per (shape, dtype,
    device, ring degree): the side stream is ordered behind the compute stream at the start of every relay and
    the compute stream behind the side stream at its end (`finish`), so a slot is never rewritten while a
    kernel of the previous call still reads it.  On host tensors (gloo tests) the same runs inline.

    Use as a context manager: `finish` must run on every exit path (it re-joins the side stream).
    """

    _SLOTS = OrderedDict()   # (shape, dtype, device, P, rank) -> [(k_slot, v_slot)] * (P-1), device tensors only

    def __init__(self, process_group, k: torch.Tensor, v: torch.Tensor):
        self.P = group_info(dist, process_group)[0]
        if self.P > 1:
            # point-to-point transfers need contiguous buffers (the reference makes K/V contiguous at
            # zigzag_ring_flash_attn.py:208-209); views stay views at ring degree 1
            k, v = k.contiguous(), v.contiguous()
        self.slots: List[Tuple[torch.Tensor, torch.Tensor]] = [(k, v)]
        self.events = [None]
        self._stream = None
        self._pending = None
        if self.P == 1:
            return
        if k.is_cuda:
            self._main = torch.cuda.current_stream()
            self._stream = _side_stream(k.device, "ring")
            self._stream.wait_stream(self._main)          # k, v are produced on the compute stream
        # Posted NOW, in front of the caller's step-0 kernels: a transfer kernel that is queued before an attention
        # launch is resident at once, one that arrives while the launch already holds every CU waits for the first
        # workgroups to drain (~0.2 ms on the causal step 0) -- and on the link-bound configs the wire is the
        # critical path.  It is ONE grouped call (direct mode), so the compute stream is not kept waiting for long.
        self._pending = (process_group, k, v)
        self.post()

    def post(self):
        """Post the transfers (idempotent; the constructor does it).  Traffic a caller puts on the ring's communicator
        itself (the travelling dK/dV) must come behind this: the communicator's internal stream runs its calls in
        order, and that traffic waits for step 0's kernels."""
        if self._pending is None:
            return
        process_group, k, v = self._pending
        self._pending = None
        cuda = k.is_cuda
        recv = self._recv_slots(k, v, dist.get_rank(process_group))
        ctx = torch.cuda.stream(self._stream) if cuda else _NullCtx()
        if kv_relay_mode(self.P) == "direct":
            with ctx:
                r = dist.get_rank(process_group)
                to_global = lambda i: dist.get_global_rank(process_group, i % self.P) if process_group is not None else i % self.P
                comm = RingComm(process_group)          # one grouped send/recv to and from every peer
                for s in range(1, self.P):
                    nk, nv = recv[s - 1]
                    dst, src = to_global(r + s), to_global(r - s)
                    comm._ops += [dist.P2POp(dist.isend, k, dst, group=process_group),
                                  dist.P2POp(dist.irecv, nk, src, group=process_group),
                                  dist.P2POp(dist.isend, v, dst, group=process_group),
                                  dist.P2POp(dist.irecv, nv, src, group=process_group)]
                    self.slots.append((nk, nv))
                comm.commit()
                comm.wait()
                ev = None
                if cuda:
                    ev = torch.cuda.Event()
                    ev.record(self._stream)
                self.events += [ev] * (self.P - 1)
            return
        with ctx:
            cur_k, cur_v = k, v
            for s in range(1, self.P):
                comm = RingComm(process_group)
                nk = comm.send_recv(cur_k, recv[s - 1][0])
                nv = comm.send_recv(cur_v, recv[s - 1][1])
                comm.commit()
                comm.wait()
                ev = None
                if cuda:
                    ev = torch.cuda.Event()
                    ev.record(self._stream)
                self.slots.append((nk, nv))
                self.events.append(ev)
                cur_k, cur_v = nk, nv

    def _recv_slots(self, k, v, rank):
        if not k.is_cuda:
            return [(torch.empty_like(k), torch.empty_like(v)) for _ in range(self.P - 1)]
        # the compute stream is part of the key: the no-rewrite guarantee (side stream ordered behind the compute stream
        # at the start of a relay, the compute stream behind the side stream at its end) holds between calls on ONE stream
        key = (tuple(k.shape), tuple(v.shape), k.dtype, k.device.index, self.P, rank,
               torch.cuda.current_stream().cuda_stream)
        return _cached_slots(KVRelay._SLOTS, key,
                             lambda: [(torch.empty_like(k), torch.empty_like(v)) for _ in range(self.P - 1)])

    def get(self, step: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """K, V held after `step` hops; makes the current stream wait for that hop."""
        ev = self.events[step]
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        return self.slots[step]

    def finish(self):
        """Order the side stream before anything the compute stream does next, so the receive slots (and
        buffers handed back to the caching allocator) cannot be reused while a hop still reads or writes them."""
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
            self._stream = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.finish()
        return False


class ZigzagKVFetch:
    """K/V transport of the zigzag FORWARD for ring degree > 2: the direct mesh fetch of KVRelay, in waves and
    without the bytes the schedule never reads.

    With the local sequence = [chunk r | chunk 2P-1-r], ring step s of rank r reads the K/V of rank r-s: only its
    FRONT half rows when s <= r, both halves (against the back half of q) when s > r
    (zigzag_ring_flash_attn.py:54-67).  Every half is cut into `pieces` row ranges; wave w (one grouped send/recv
    each, 2 * pieces waves) carries piece w % pieces of half w // pieces:
      front waves: every rank sends the piece to all P-1 peers;
      back waves:  the piece goes only to the peers that read it (destination rank (r+s) mod P with r+s >= P),
                   a quarter of all K/V bytes is never sent;
    and the attention of a step is one launch per piece it reads, merged by the kernel's fused LSE merge, issued
    WAVE by wave (zigzag_ring_flash_attn_forward): everything that needs only the waves that have landed runs
    before the compute stream waits for the next one.  Where the ring is link-bound (BASELINE's 4-GPU config:
    64 MiB of K/V per peer = ~1 ms per xGMI link against ~1 ms of attention per rank) what waits for the wire
    shrinks from three ring steps to the launches of the last piece.  All waves run on the "ring" side stream;
    receive slots are persistent like KVRelay's.  Context manager, like KVRelay."""

    _SLOTS = OrderedDict()

    def __init__(self, process_group, k: torch.Tensor, v: torch.Tensor, pieces: int = 1):
        P = self.P = dist.get_world_size(process_group)
        r = self.r = dist.get_rank(process_group)
        assert P > 2 and k.shape[1] % 2 == 0
        c = k.shape[1] // 2
        W = self.pieces = max(1, min(int(pieces), c))
        self.waves = 2 * W
        cut = [(i * c) // W for i in range(W + 1)]
        # mine[w] = [k piece, v piece] of wave w, contiguous (views at B = 1)
        mine = [[t[:, h * c + cut[i]:h * c + cut[i + 1]].contiguous() for t in (k, v)] for h in (0, 1) for i in range(W)]
        cuda = k.is_cuda
        self._stream = None
        if cuda:
            self._stream = _side_stream(k.device, "ring")
            self._stream.wait_stream(torch.cuda.current_stream())     # k, v are produced on the compute stream
        # One buffer pair per wave: the pieces of ring ranks r-1 ... r-(P-1) one behind the other along the sequence.
        # At batch 1 each piece is a contiguous row range of it (what a receive needs), so the pieces of several
        # source ranks can be handed to ONE attention launch as one K/V (`get_range`): one merge epilogue per wave and
        # query range instead of one per source rank (+16 / +40 us each at BASELINE's 4-GPU shape, kbench pieces).
        self.grouped = k.shape[0] == 1 and os.environ.get("USP_ZZ_GROUP", "1") != "0"      # USP_ZZ_GROUP=0: A/B switch
        key = (tuple(k.shape), tuple(v.shape), k.dtype, k.device.index if cuda else -1, P, r, W, self.grouped,
               torch.cuda.current_stream().cuda_stream if cuda else 0)      # per compute stream, as KVRelay's

        def make():
            if not self.grouped:
                return None, [[tuple(torch.empty_like(t) for t in mine[w]) for _ in range(P - 1)] for w in range(2 * W)]
            bufs = [tuple(torch.empty((1, (P - 1) * t.shape[1]) + tuple(t.shape[2:]), dtype=t.dtype, device=t.device)
                          for t in mine[w]) for w in range(2 * W)]
            rows = [mine[w][0].shape[1] for w in range(2 * W)]
            return bufs, [[tuple(b[:, i * rows[w]:(i + 1) * rows[w]] for b in bufs[w]) for i in range(P - 1)]
                          for w in range(2 * W)]
        # slots[w][s-1] = (k, v) piece of rank r-s (views into bufs[w] when grouped)
        self.bufs, self.slots = _cached_slots(ZigzagKVFetch._SLOTS, key, make) if cuda else make()
        self.events = [None] * (2 * W)
        # Wave 0 is posted NOW, in front of the caller's step-0 kernels (a transfer kernel queued before an attention
        # launch is resident at once; one that arrives while the launch holds every CU waits for workgroups to drain,
        # and on the link-bound configs the wire is the critical path); the other 2 W - 1 grouped calls are posted by
        # the first get(), i.e. BEHIND the launch of step 0, so the compute stream has work while the host walks
        # through them (they queue behind wave 0 on the communicator's stream anyway).
        self._pending = (process_group, mine)
        self._posted = 0
        self.post(1)

    def post(self, upto=None):
        """Post waves [posted, upto) (all that are left when upto is None); idempotent."""
        upto = 2 * self.pieces if upto is None else upto
        if self._pending is None or self._posted >= upto:
            return
        process_group, mine = self._pending
        first, self._posted = self._posted, upto
        if upto == 2 * self.pieces:
            self._pending = None
        P, r, W = self.P, self.r, self.pieces
        slots, cuda = self.slots, self._stream is not None
        to_global = lambda i: dist.get_global_rank(process_group, i % P) if process_group is not None else i % P
        with (torch.cuda.stream(self._stream) if cuda else _NullCtx()):
            for w in range(first, upto):
                send_steps, recv_steps = zigzag_wave_steps(P, r, w < W)
                comm = RingComm(process_group)
                for s in range(1, P):
                    if s in send_steps:
                        comm._ops += [dist.P2POp(dist.isend, t, to_global(r + s), group=process_group) for t in mine[w]]
                    if s in recv_steps:
                        comm._ops += [dist.P2POp(dist.irecv, t, to_global(r - s), group=process_group)
                                      for t in slots[w][s - 1]]
                comm.commit()
                comm.wait()
                if cuda:
                    self.events[w] = torch.cuda.Event()
                    self.events[w].record(self._stream)
                    for t in mine[w]:
                        t.record_stream(self._stream)

    def get(self, wave: int, step: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(K, V) rows of wave `wave` (piece wave % pieces of the front half for wave < pieces, of the back half
        otherwise) of ring rank r - step; the compute stream waits for that wave."""
        assert wave < self.pieces or step > self.r, "the zigzag schedule never reads this half"
        self.post()
        if self.events[wave] is not None:
            torch.cuda.current_stream().wait_event(self.events[wave])
        return self.slots[wave][step - 1]

    def get_range(self, wave: int, s_lo: int, s_hi: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(K, V) rows of wave `wave` of ring ranks r - s_lo ... r - s_hi as ONE tensor each (batch 1 only: `grouped`);
        the compute stream waits for that wave."""
        assert self.grouped and 1 <= s_lo <= s_hi < self.P
        assert wave < self.pieces or s_lo > self.r, "the zigzag schedule never reads this half"
        self.post()
        if self.events[wave] is not None:
            torch.cuda.current_stream().wait_event(self.events[wave])
        rows = self.slots[wave][0][0].shape[1]
        return tuple(b[:, (s_lo - 1) * rows:s_hi * rows] for b in self.bufs[wave])

    def finish(self):
        # every rank issues the same sequence of grouped calls on every exit path: a rank that leaves before its first
        # get() (an exception in the step-0 launch, a plan without launches) must still post its sends, or its peers
        # hang in RCCL instead of failing
        self.post()
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
            self._stream = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.finish()
        return False


def clear_slot_caches():
    """Drop the persistent K/V receive slots (they return to the caching allocator): for callers that change shapes for
    good, e.g. between a long-sequence phase and a packed-batch phase."""
    KVRelay._SLOTS.clear()
    ZigzagKVFetch._SLOTS.clear()


def zigzag_wave_steps(P: int, r: int, front: bool):
    """(steps s whose destination rank (r+s) % P gets this wave from rank r, steps s whose source rank (r-s) % P
    sends this wave to rank r).  Front-half waves go to everyone; a back-half wave goes to rank (r+s) % P only if that
    rank reads it, i.e. its step s is beyond its rank: r + s >= P; and rank r reads the back half of source r - s
    when s > r (zigzag_ring_flash_attn.py:59-67)."""
    steps = range(1, P)
    if front:
        return set(steps), set(steps)
    return {s for s in steps if r + s >= P}, {s for s in steps if s > r}


def zigzag_fetch_pieces(k: torch.Tensor) -> int:
    """Row ranges per K/V half of the zigzag mesh fetch: USP_ZZ_PIECES, default 2 where a half of one peer's K + V is
    at least 16 MiB (a quarter of a millisecond on an xGMI link: BASELINE's 4-GPU config moves 32 MiB per half against
    0.14 ms of kernels per half step), else 1 (the 8-GPU config moves 4 MiB per half: more launches would buy nothing)."""
    env = os.environ.get("USP_ZZ_PIECES")
    if env:
        return max(1, int(env))
    half_bytes = k.numel() * k.element_size()                   # half the rows of K and of V
    return 2 if half_bytes >= (16 << 20) else 1


def kv_relay_mode(P: int) -> str:
    """"direct" (mesh fetch) or "chain" (hop-by-hop relay): USP_KV_RELAY, default direct for ring degree > 2."""
    return os.environ.get("USP_KV_RELAY", "direct" if P > 2 else "chain")


def dkdv_return_mode() -> str:
    """"relay" (the reference's hop-by-hop travel, default) or "direct" (every block straight to its owner):
    USP_DKDV_RETURN."""
    return os.environ.get("USP_DKDV_RETURN", "relay")


FULL = slice(None)


def last_hop_16bit() -> bool:
    """USP_DKDV_LAST_HOP=fp32 keeps the reference's fp32 payload on the last hop of the travelling dK/dV (A/B switch)."""
    return os.environ.get("USP_DKDV_LAST_HOP", "16") != "fp32"


def travel_dkdv(process_group, k, v, block, fold, zero: bool = False, extent=None, be=None, final_dtype=None,
                defer=None, split_block: bool = False):
    """The travelling dK/dV of every ring backward (zigzag_ring_flash_attn.py:139-183, ring_flash_attn.py:
    86-147): the fp32 accumulators of K/V block j visit every rank that attends to it, one hop per step,
    each rank adding its block result; after P hops they are back home.

        block(step, kk, vv, dk_dst, dv_dst) -> False if the step computes nothing
            runs the block backward of `step` against the K/V that arrived after `step` hops, writing the
            dK/dV block into dk_dst/dv_dst (at step 0 these ARE the travelling accumulators);
        fold(step, dk_acc, dv_acc, dk_blk, dv_blk)
            adds this step's block into the accumulators that just arrived;
        extent(rank, step) -> None | slice
            the K/V rows (dim 1) the block of `step` on ring rank `rank` carries gradients for (None: that step
            computes nothing there; FULL: all rows).  Given together with `be`, USP_DKDV_RETURN=direct is honoured
            (`return_dkdv_direct`).
        final_dtype (with `be`): the LAST hop carries the accumulators already rounded to that 16-bit type.  Nothing is
            added after the last hop -- the owner only rounds what arrives (the reference: `dk.to(q.dtype)`,
            zigzag_ring_flash_attn.py:181-183) -- so rounding in front of the hop is bit-identical and halves the
            one hop no kernel can hide (it follows the last step's kernels): 16 -> 8 MiB per KV head at BASELINE's
            8-GPU config.  The function then returns the 16-bit tensors (final_grads passes them through).
        defer: a list.  The wait for the last hop is NOT queued on the calling stream; the pending RingComm is appended
            to `defer` and whoever consumes dK/dV calls its wait() on the consuming stream (the head-group pipeline
            consumes it on the exchange lane, so the next group's kernels start behind this group's last kernels, not
            behind its last hop).

        split_block: `block` takes `only=` ("dkdv" | "dq") and is called TWICE per step: the dK/dV launch, then -- behind the
            fold and the posting of the step's hop -- the dQ launch.  The hop then runs beside the step's own dQ launch as well,
            and the LAST hop, which nothing used to hide on a ring-only grid (the 4-GPU grid: 32 MiB = 0.5 ms at 64 GB/s of a
            30 ms iteration), runs beside the last step's dQ launch.  Same kernels, same sums, same order: bit-identical.

    The hop of step s is posted from the compute stream right after the kernels of step s (split_block: after its dK/dV
    launch), and runs beside the kernels of step s+1.  `zero`: start every buffer from zeros (packed batches: the kernels do not touch
    rows outside every sequence's range, which would otherwise travel -- and be summed -- uninitialised).
    Returns the final (dk, dv): fp32 accumulators, or `final_dtype` tensors."""
    P = group_info(dist, process_group)[0]
    if P > 1 and extent is not None and be is not None and dkdv_return_mode() == "direct":
        return return_dkdv_direct(process_group, k, v, block, extent, be, zero)
    new = (lambda t: torch.zeros(t.shape, dtype=torch.float32, device=t.device)) if zero else \
          (lambda t: torch.empty(t.shape, dtype=torch.float32, device=t.device))
    round_last = P > 1 and be is not None and final_dtype is not None and final_dtype != torch.float32 and last_hop_16bit()
    dk_blk, dv_blk = new(k), new(v)
    d_comm = None
    dk_acc = dv_acc = next_dk = next_dv = None
    with KVRelay(process_group, k, v) as relay:
        for step in range(P):
            kk, vv = relay.get(step)
            bkw = {"only": "dkdv"} if split_block else {}
            if step == 0:
                dk_acc, dv_acc = new(k), new(v)
                block(0, kk, vv, dk_acc, dv_acc, **bkw)
            else:
                computed = block(step, kk, vv, dk_blk, dv_blk, **bkw)
                d_comm.wait()                       # the travelling accumulators of step-1 have landed
                dk_acc, dv_acc = next_dk, next_dv
                if computed is not False:
                    fold(step, dk_acc, dv_acc, dk_blk, dv_blk)
            if round_last and step == P - 1:        # the hop home: rounded here instead of at its destination
                dk_acc, dv_acc = (_rounded(be, t, final_dtype) for t in (dk_acc, dv_acc))
            d_comm = RingComm(process_group)
            next_dk = d_comm.send_recv(dk_acc)
            next_dv = d_comm.send_recv(dv_acc)
            d_comm.commit()
            if split_block:                         # the step's dQ launch, beside its own hop
                block(step, kk, vv, None, None, only="dq")
        if defer is None or not round_last:      # (an fp32 arrival still has to be rounded on this stream)
            d_comm.wait()
        else:
            d_comm.keep = (dk_acc, dv_acc)          # the send buffers live until the consumer has waited
            defer.append(d_comm)
    return next_dk, next_dv


def _rounded(be, acc, dtype):
    out = torch.empty(acc.shape, dtype=dtype, device=acc.device)
    be.cast(out, acc)
    return out


def return_dkdv_direct(process_group, k, v, block, extent, be, zero: bool = False):
    """dK/dV without the relay (USP_DKDV_RETURN=direct): ring rank r computes at step s the block of the K/V owned
    by rank r-s and sends it STRAIGHT to that owner, which adds the P-1 arriving blocks to its own step-0 block in
    step order -- the order the relay adds them in, so the result is bit-identical to it.

    What the xGMI mesh buys: the relay moves the whole fp32 accumulator over the ONE link to rank r+1 every step,
    and hop s cannot start before hop s-1 has landed AND step s has been computed; here the block of step s rides
    the link to rank r-s (a different link every step), depends on nothing but its own kernels, and carries only the
    rows it has gradients for (`extent`: the zigzag steps s <= r produce front-half rows only, a quarter of all bytes
    is never sent).  At the 8-GPU BASELINE config either form hides behind 3.4 ms of kernels per step; an MHA ring
    (2 x 64 MiB of fp32 per hop at ring 4 x 8192 tokens x 16 heads, ~2 ms per link against ~0.9 ms of kernels) is
    link-bound in the relay and not here.  Costs P-1 receive buffers instead of one (sized for 288 GB).
    Every step's send is posted from the compute stream behind that step's kernels, like the relay's hops."""
    P = dist.get_world_size(process_group)
    r = dist.get_rank(process_group)
    new = (lambda shape, dev: torch.zeros(shape, dtype=torch.float32, device=dev)) if zero else \
          (lambda shape, dev: torch.empty(shape, dtype=torch.float32, device=dev))
    to_global = (lambda i: dist.get_global_rank(process_group, i % P)) if process_group is not None else (lambda i: i % P)

    def rows(t, sl):                                    # the rows of a block that carry gradients
        return t if sl == FULL else t[:, sl]

    def wire(t, sl):                                    # ... as a contiguous tensor (a view at batch 1)
        return rows(t, sl).contiguous()

    pending = []
    with KVRelay(process_group, k, v) as relay:
        kk, vv = relay.get(0)
        dk_acc, dv_acc = new(k.shape, k.device), new(v.shape, v.device)
        block(0, kk, vv, dk_acc, dv_acc)
        for step in range(1, P):
            kk, vv = relay.get(step)
            out_sl, in_sl = extent(r, step), extent((r + step) % P, step)
            comm = RingComm(process_group)
            keep = None
            if out_sl is not None:
                dk_blk, dv_blk = new(k.shape, k.device), new(v.shape, v.device)      # one pair per step: the send
                block(step, kk, vv, dk_blk, dv_blk)                                  # reads it beside later steps
                keep = (wire(dk_blk, out_sl), wire(dv_blk, out_sl))
                comm._ops += [dist.P2POp(dist.isend, t, to_global(r - step), group=process_group) for t in keep]
            got = None
            if in_sl is not None:
                got = tuple(new(rows(t, in_sl).shape, t.device) for t in (k, v))
                comm._ops += [dist.P2POp(dist.irecv, t, to_global(r + step), group=process_group) for t in got]
            if comm._ops:
                comm.commit()
                pending.append((comm, in_sl, got, keep))
        for comm, in_sl, got, _keep in pending:         # step order == the relay's summation order
            comm.wait()
            if got is not None:
                be.add(rows(dk_acc, in_sl), rows(dk_acc, in_sl), got[0])
                be.add(rows(dv_acc, in_sl), rows(dv_acc, in_sl), got[1])
    return dk_acc, dv_acc


def final_grads(be, refs, accs):
    """16-bit gradients from the fp32 accumulators of a ring backward: fresh CONTIGUOUS tensors (the cast
    kernel takes rows; `empty_like` would inherit the seq-major strides the Ulysses exchange hands the ring
    for batch > 1).  Accumulators that arrived rounded (travel_dkdv's `final_dtype`) pass through."""
    out = []
    for ref, acc in zip(refs, accs):
        if acc.dtype == ref.dtype:
            out.append(acc)
            continue
        g = torch.empty(ref.shape, dtype=ref.dtype, device=ref.device)
        be.cast(g, acc)
        out.append(g)
    return tuple(out)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_SIDE_STREAMS = {}


def _side_stream(device, lane: str = "ring") -> "torch.cuda.Stream":
    """One side HIP stream per device and lane: "ring" carries the K/V relay, "ulysses" the pipelined head
    exchange -- two lanes, so a ring hop never queues behind the exchanges of later head groups."""
    key = (device.type, device.index, lane)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]
