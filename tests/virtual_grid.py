"""A ulysses x ring GRID of virtual ranks inside one process (threads): a torch.distributed stand-in for the modules of
the package, so that the real layer code -- packed q|k|v exchange pipelined over head groups, ring schedules, relay /
mesh fetch, travelling dK/dV -- runs for every rank of a grid without one process per rank.

Two transports:
  * device tensors (tests/test_gpu_rccl_order.py): every point-to-point batch of a ring group and every
    all_to_all_single of a ulysses group becomes ONE grouped RCCL self send/recv call over the members' buffers on a
    real 1-rank NCCL group, issued behind every member's current stream and waited for by every member's current stream:
    real RCCL, stream-ordered only, no host synchronisation of the device;
  * host tensors (tests/test_dist_cpu.py): plain copies -- the same harness checks the grid logic without a GPU.
The virtual ranks meet on the HOST at every collective (a barrier per group), which real ranks do not; what is real is
the device-side ordering between compute streams, side streams and the transfers."""
import threading

import torch
import torch.distributed


class Group:
    """One process group of the virtual grid (what the schedules pass around as `group` / `process_group`)."""

    def __init__(self, kind, members):
        self.kind, self.members = kind, list(members)          # global ranks, in group-rank order
        self.barrier = threading.Barrier(len(members), timeout=180)
        self.pending = {}                                       # global rank -> (payload, ready event)
        self.done = None


class Req:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):                                             # stream-wise, like ProcessGroupNCCL's work.wait()
        if self.ev is not None:
            torch.cuda.current_stream().wait_event(self.ev)


class Ctx:
    """What the layer's autograd Function asks of its context (the tests drive forward / backward by hand: autograd
    would run every virtual rank's backward on the ONE worker thread it keeps per device, where they cannot wait for
    each other)."""
    needs_input_grad = (True, True, True)
    saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


class VirtualGrid:
    """use_ulysses_low layout (globals.py:39-57: ulysses ranks contiguous, ring ranks strided)."""

    isend, irecv = "send", "recv"

    def __init__(self, ud, rd, real_dist=None):
        self.ud, self.rd, self.d = ud, rd, real_dist            # real_dist None: host tensors, plain copies
        self.tls = threading.local()
        self.ulysses = [Group("ulysses", [r * ud + u for u in range(ud)]) for r in range(rd)]
        self.ring = [Group("ring", [r * ud + u for r in range(rd)]) for u in range(ud)]
        self.world = Group("world", range(ud * rd))             # group=None in a P2POp: the relayed pair exchange
        self.calls = []                                         # (kind, first member) in issue order
        self.issue = threading.Lock()                           # torch's coalescing manager keeps per-group global state:
                                                                # one real grouped call at a time

    def groups_of(self, rank):
        return self.ulysses[rank // self.ud], self.ring[rank % self.ud]

    def abort(self):
        for g in self.ulysses + self.ring + [self.world]:
            g.barrier.abort()

    # --- queries ---------------------------------------------------------------------------------------------
    def get_world_size(self, group=None):
        group = self.world if group is None else group
        return len(group.members) if isinstance(group, Group) else self.d.get_world_size(group)

    def get_rank(self, group=None):
        group = self.world if group is None else group
        return group.members.index(self.tls.rank) if isinstance(group, Group) else self.d.get_rank(group)

    def get_global_rank(self, group, r):
        return group.members[r] if isinstance(group, Group) else self.d.get_global_rank(group, r)

    def P2POp(self, op, tensor, peer, group=None):
        return (op, tensor, peer, self.world if group is None else group)

    # --- the two kinds of traffic ----------------------------------------------------------------------------
    def _joint(self, group, payload, pairs_of):
        """Every member calls this with its payload; ONE thread turns all payloads into (source, destination) tensor
        pairs and issues them as one grouped call on ITS current stream, behind every member's current stream; the
        returned event is what every member's stream has to wait for (None on host tensors)."""
        me = self.tls.rank
        cuda = self.d is not None
        ready = None
        if cuda:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
        group.pending[me] = (payload, ready)
        if group.barrier.wait() == 0:
            pairs = pairs_of({m: group.pending[m][0] for m in group.members})
            if cuda:
                cur = torch.cuda.current_stream()
                for _, ev in group.pending.values():
                    cur.wait_event(ev)
                ops = []
                for src, dst in pairs:
                    ops += [self.d.P2POp(self.d.isend, src, 0), self.d.P2POp(self.d.irecv, dst, 0)]
                with self.issue:
                    reqs = self.d.batch_isend_irecv(ops)
                for req in reqs:
                    req.wait()
                group.done = torch.cuda.Event()
                group.done.record(cur)
            else:
                staged = [(dst, src.clone()) for src, dst in pairs]     # a send buffer may be another pair's destination
                for dst, val in staged:
                    dst.copy_(val)
                group.done = None
            self.calls.append((group.kind, group.members[0]))
        group.barrier.wait()
        done = group.done
        group.barrier.wait()                                    # everyone holds `done` before it is replaced
        return done

    def batch_isend_irecv(self, ops):
        group = ops[0][3]
        assert isinstance(group, Group) and group.kind in ("ring", "world") and all(o[3] is group for o in ops)

        def pairs_of(per_rank):
            pairs = []
            for src in group.members:                           # message order = (source rank, its send order)
                nth = {}
                for _, tensor, dst, _g in (o for o in per_rank[src] if o[0] == "send"):
                    j = nth.get(dst, 0)
                    nth[dst] = j + 1
                    recvs = [o for o in per_rank[dst] if o[0] == "recv" and o[2] == src]
                    assert recvs[j][1].shape == tensor.shape, (src, dst, j, recvs[j][1].shape, tensor.shape)
                    pairs.append((tensor, recvs[j][1]))
            n_recv = sum(1 for m in group.members for o in per_rank[m] if o[0] == "recv")
            assert n_recv == len(pairs), "unmatched receives in a grouped call"
            return pairs
        return [Req(self._joint(group, ops, pairs_of))]

    ReduceOp = torch.distributed.ReduceOp

    def all_reduce(self, t, op=None, group=None):
        """MAX / MIN / SUM over the members of a virtual group (the relayed exchange agrees on its buffer shape once;
        comm/relay_exchange.py:_agree_once): values meet on the host, every member gets the result."""
        group = self.world if group is None else group
        assert isinstance(group, Group)
        group.pending[self.tls.rank] = t.detach().to("cpu", copy=True)
        group.barrier.wait()
        vals = torch.stack([group.pending[m] for m in group.members])
        red = {self.ReduceOp.MAX: vals.max(0).values, self.ReduceOp.MIN: vals.min(0).values}.get(op, vals.sum(0))
        group.barrier.wait()                                    # everyone has read `pending` before it is reused
        t.copy_(red.to(t.device))

    def all_to_all_single(self, recv, send, group=None):
        assert isinstance(group, Group) and group.kind == "ulysses"
        P = len(group.members)

        def pairs_of(per_rank):                                 # chunk d of member s's send -> chunk s of member d's recv
            return [(per_rank[s][0].chunk(P)[di], per_rank[d][1].chunk(P)[si])
                    for si, s in enumerate(group.members) for di, d in enumerate(group.members)]
        done = self._joint(group, (send, recv), pairs_of)
        if done is not None:
            torch.cuda.current_stream().wait_event(done)


def patch_dist(monkeypatch, grid):
    """Point every module of the package that talks to torch.distributed at the virtual grid."""
    import yunchang_amd.comm.all_to_all as A
    import yunchang_amd.comm.relay_exchange as RX
    import yunchang_amd.hybrid.async_attn_layer as AL
    import yunchang_amd.ring.ring_flash_attn as R
    import yunchang_amd.ring.stripe_flash_attn as S
    import yunchang_amd.ring.utils as U
    import yunchang_amd.ring.zigzag_ring_flash_attn as Z
    for mod in (U, Z, R, S, AL, A, RX):
        monkeypatch.setattr(mod, "dist", grid)
    monkeypatch.setattr(RX, "GRID", (grid.ud, grid.rd, grid.ud * grid.rd, True))
    return AL


def run_grid(grid, ws, rank_fn):
    """rank_fn(rank) on `ws` threads; returns the per-rank results (raises the first error)."""
    res, errs = [None] * ws, []

    def body(r):
        try:
            grid.tls.rank = r
            res[r] = rank_fn(r)
        except BaseException as e:                              # noqa: BLE001 - report, and release the others
            import traceback
            errs.append((r, repr(e), traceback.format_exc()))
            grid.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(ws)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errs, errs[0]
    return res


class VirtualGridPairwise(VirtualGrid):
    """The same grid WITHOUT a host rendezvous of the whole group at every collective: every message is matched between
    its two ranks only (a mailbox per ordered rank pair), so the virtual ranks drift apart on the host the way real ranks
    do -- a rank may be a whole ring step ahead of its neighbour -- and a schedule whose ranks post their transfers in
    different orders deadlocks here (a timeout) instead of being straightened out by a group barrier.  The receiver issues
    each transfer (one grouped RCCL self send/recv, behind the sender's and its own stream) and hands the completion event
    back; a sender's wait() blocks on the host only until its receiver has issued.  `jitter` = (seed, max seconds): random
    host delays in front of every posting, to shake the interleavings."""

    def __init__(self, ud, rd, real_dist=None, jitter=None):
        super().__init__(ud, rd, real_dist)
        import collections
        import queue
        self.box = collections.defaultdict(queue.Queue)          # (src, dst) -> messages in posting order
        self.jitter = jitter
        self.aborted = threading.Event()

    def abort(self):
        self.aborted.set()
        super().abort()

    def _nap(self):
        if self.jitter:
            import random
            import time
            rnd = self.tls.__dict__.setdefault("rnd", random.Random(self.jitter[0] * 1000 + self.tls.rank))
            time.sleep(rnd.random() * self.jitter[1])

    def _post(self, tensor, dst):
        ready = None
        if self.d is not None:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
        msg = {"t": tensor, "ready": ready, "done": threading.Event(), "ev": None}
        self.box[(self.tls.rank, dst)].put(msg)
        return msg

    def _take(self, tensor, src):
        import queue
        while True:                                              # the sender has not posted yet: wait (bounded)
            try:
                msg = self.box[(src, self.tls.rank)].get(timeout=1.0)
                break
            except queue.Empty:
                if self.aborted.is_set():
                    raise RuntimeError("virtual grid aborted")
                self._waited = getattr(self, "_waited", 0) + 1
                if self._waited > 900:        # (cumulative over the grid's rank threads: minutes of real waiting, i.e. a deadlock, not a loaded host)
                    raise TimeoutError(f"rank {self.tls.rank}: no message from rank {src} -- the ranks' posting orders differ")
        assert msg["t"].shape == tensor.shape and msg["t"].dtype == tensor.dtype, (src, self.tls.rank, msg["t"].shape, tensor.shape)
        if self.d is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(msg["ready"])
            with self.issue:
                reqs = self.d.batch_isend_irecv([self.d.P2POp(self.d.isend, msg["t"], 0), self.d.P2POp(self.d.irecv, tensor, 0)])
            for req in reqs:
                req.wait()
            msg["ev"] = torch.cuda.Event()
            msg["ev"].record(cur)
        else:
            tensor.copy_(msg["t"])
        msg["done"].set()
        return msg

    class _PairReq:
        def __init__(self, grid, sent, got):
            self.grid, self.sent, self.got = grid, sent, got

        def wait(self):
            for msg in self.sent:                                # (host) until the receiver has issued the transfer ...
                while not msg["done"].wait(timeout=1.0):
                    if self.grid.aborted.is_set():
                        raise RuntimeError("virtual grid aborted")
            if self.grid.d is not None:                          # ... then stream-wise, like ProcessGroupNCCL's work.wait()
                cur = torch.cuda.current_stream()
                for msg in self.sent + self.got:
                    cur.wait_event(msg["ev"])

    def batch_isend_irecv(self, ops):
        self._nap()
        kind = ops[0][3].kind
        self.calls.append((kind, self.tls.rank))
        sent = [self._post(t, dst) for op, t, dst, _ in ops if op == "send"]      # every send first, then the receives
        got = [self._take(t, src) for op, t, src, _ in ops if op == "recv"]
        return [self._PairReq(self, sent, got)]

    def all_to_all_single(self, recv, send, group=None):
        assert isinstance(group, Group) and group.kind == "ulysses"
        self._nap()
        self.calls.append(("ulysses", self.tls.rank))
        P, me = len(group.members), group.members.index(self.tls.rank)
        recv.chunk(P)[me].copy_(send.chunk(P)[me])
        sent = [self._post(send.chunk(P)[di], d) for di, d in enumerate(group.members) if di != me]
        got = [self._take(recv.chunk(P)[si], s) for si, s in enumerate(group.members) if si != me]
        self._PairReq(self, sent, got).wait()
