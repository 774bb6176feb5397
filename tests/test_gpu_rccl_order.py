"""Compute stream -> RCCL point-to-point ordering, on ONE GPU, through REAL RCCL.

The multi-process tests (test_gpu_multiproc.py) have to use gloo (RCCL refuses two ranks per device) and insert a
device synchronise in front of every RingComm.commit, which hides exactly the ordering that matters on real
hardware: a send posted right behind the kernel that produces its payload (the travelling fp32 dK/dV accumulators
of the ring backward, ring/utils.py:travel_dkdv) and a kernel launched right behind the receive it consumes.

Here a ring of P VIRTUAL ranks lives in one process: every virtual rank is a thread with its own HIP stream that
runs the package's real zigzag ring forward / backward (HIP kernels, KVRelay in chain mode on the side stream,
RingComm); the wire is a 1-rank NCCL group, i.e. RCCL self send/recv (each hop's P sends and P receives are one
grouped call, matched first-in first-out).  No host synchronisation anywhere: a rank's sends are ordered behind
its compute by stream events, as ProcessGroupNCCL orders its internal stream behind the caller's current stream,
and its next kernels behind the receive.  Every iteration must be bit-identical to the first and match the reference golden."""
import os
import threading

import numpy as np
import pytest
import torch

from golden_util import Golden, TOL, assert_close, golden_files

pytestmark = pytest.mark.gpu


class _VirtualRing:
    """torch.distributed stand-in for P virtual ranks (threads) of one ring, backed by a real 1-rank NCCL group."""

    def __init__(self, P, real_dist):
        self.P, self.d = P, real_dist
        self.tls = threading.local()
        self.barrier = threading.Barrier(P, timeout=120)
        self.pending = [None] * P            # per rank: (ops, event recorded on the rank's current stream)
        self.done = None                     # event: the grouped call of this round has completed
        self.group = object()                # the handle the schedules pass around as `process_group`

    # --- what ring/utils.py and the ring schedules ask torch.distributed -------------------------------------
    def get_world_size(self, group=None):
        return self.P if group is self.group else self.d.get_world_size(group)

    def get_rank(self, group=None):
        return self.tls.rank if group is self.group else self.d.get_rank(group)

    def get_global_rank(self, group, r):
        return r if group is self.group else self.d.get_global_rank(group, r)

    def P2POp(self, op, tensor, peer, group=None):
        assert op in ("send", "recv")
        return (op, tensor, peer)

    isend, irecv = "send", "recv"            # markers: RingComm only ever hands them to P2POp

    def batch_isend_irecv(self, ops):
        me = self.tls.rank
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())            # this rank's payloads are produced by here
        self.pending[me] = (ops, ready)
        i = self.barrier.wait()
        if i == 0:                                            # one thread issues the grouped call for everyone
            real_ops = []
            for src in range(self.P):                         # message order = (source rank, its send order)
                sends = [o for o in self.pending[src][0] if o[0] == "send"]
                nth = {}
                for _, tensor, dst in sends:
                    j = nth.get(dst, 0); nth[dst] = j + 1
                    recvs = [o for o in self.pending[dst][0] if o[0] == "recv" and o[2] == src]
                    real_ops.append(self.d.P2POp(self.d.isend, tensor, 0))
                    real_ops.append(self.d.P2POp(self.d.irecv, recvs[j][1], 0))
            cur = torch.cuda.current_stream()
            for _, ev in self.pending:                        # RCCL runs behind EVERY rank's compute ...
                cur.wait_event(ev)
            for req in self.d.batch_isend_irecv(real_ops):
                req.wait()                                    # stream-wise: `cur` waits for the transfer
            self.done = torch.cuda.Event()
            self.done.record(cur)
        self.barrier.wait()
        done = self.done
        self.barrier.wait()                                   # everyone has picked up `done` before it is replaced
        return [_Req(done)]


class _Req:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):                                           # ... and every rank's next kernels behind the transfer
        torch.cuda.current_stream().wait_event(self.ev)


@pytest.fixture(scope="module")
def nccl_single():
    import torch.distributed as dist
    import yunchang_amd  # noqa: F401
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29733")
    own = not dist.is_initialized()
    if own:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)
    yield dist
    if own:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("relay,iters", [("chain", 50), ("direct", 20)])
def test_ring_backward_ordering_through_rccl_self_sendrecv(nccl_single, monkeypatch, relay, iters):
    import yunchang_amd.ring.utils as U
    import yunchang_amd.ring.zigzag_ring_flash_attn as Z
    real = nccl_single
    dev = torch.device("cuda:0")
    # can this RCCL build send to itself at all?
    a, b = torch.arange(8, device=dev, dtype=torch.float32), torch.zeros(8, device=dev)
    try:
        for req in real.batch_isend_irecv([real.P2POp(real.isend, a, 0), real.P2POp(real.irecv, b, 0)]):
            req.wait()
        torch.cuda.synchronize()
    except Exception as e:                                    # pragma: no cover - depends on the RCCL build
        pytest.skip(f"RCCL self send/recv not available: {e!r}")
    assert torch.equal(a, b)

    g = Golden([f for f in golden_files() if "c4_w4_u1r4" in f][0])          # ring 4, zigzag, forward + backward
    P = g.rd
    ring = _VirtualRing(P, real)
    for mod in (U, Z):
        monkeypatch.setattr(mod, "dist", ring)
    # "chain": hop-by-hop relay, one event per slot, P-1 grouped calls.  "direct" (the default at ring degree 4): the
    # forward's two-wave mesh fetch (ZigzagKVFetch: ranks post DIFFERENT numbers of sends / receives per wave) and the
    # backward's one-call mesh fetch
    monkeypatch.setenv("USP_KV_RELAY", relay)
    dtype = getattr(torch, g.dtype)
    loc = [[torch.from_numpy(np.ascontiguousarray(g.shard(x, r))).to(dtype).to(dev) for x in (g.q, g.k, g.v, g.dout)]
           for r in range(P)]
    scale = g.D ** -0.5
    streams = [torch.cuda.Stream(device=dev) for _ in range(P)]
    torch.cuda.synchronize()

    def run_rank(r, res, errs):
        try:
            ring.tls.rank = r
            torch.cuda.set_device(dev)
            q, k, v, do = loc[r]
            with torch.cuda.stream(streams[r]):
                out, lse = Z.zigzag_ring_flash_attn_forward(ring.group, q, k, v, scale)
                dq, dk, dv = Z.zigzag_ring_flash_attn_backward(ring.group, do, q, k, v, out, lse, scale)
            res[r] = (out, dq, dk, dv)
        except BaseException as e:                            # noqa: BLE001 - report, and release the others
            errs.append((r, repr(e)))
            ring.barrier.abort()

    first = None
    for it in range(iters):
        res, errs = [None] * P, []
        threads = [threading.Thread(target=run_rank, args=(r, res, errs)) for r in range(P)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(300)
        assert not errs, errs
        torch.cuda.synchronize()
        got = [[t.clone() for t in res[r]] for r in range(P)]
        if first is None:
            first = got
            for r in range(P):
                for t, name in zip(got[r], ("out", "dq", "dk", "dv")):
                    tol = TOL[g.dtype]["out" if name == "out" else "grad"]
                    assert_close(t.float().cpu().numpy(), getattr(g, name)[r], *tol, f"{g.name} {name} rank {r}")
        else:
            for r in range(P):
                for t, t0, name in zip(got[r], first[r], ("out", "dq", "dk", "dv")):
                    assert torch.equal(t, t0), f"iteration {it}: {name} of rank {r} differs from iteration 0"


# ---------------------------------------------------------------------------------------------------------------------
# A ulysses x ring GRID of virtual ranks: the pipelined Ulysses exchange ("ulysses" lane) beside the ring traffic ("ring"
# lane + the travelling dK/dV posted from the compute streams) -- TWO communicators' worth of traffic in flight, every
# transfer a real RCCL self send/recv, no host synchronisation.
# ---------------------------------------------------------------------------------------------------------------------
from virtual_grid import VirtualGrid as _VirtualGrid, Ctx as _Ctx


GRID = [f for f in golden_files() if Golden(f).ud > 1 and Golden(f).rd > 1 and Golden(f).layer == "hybrid"
        and Golden(f).dtype == "bfloat16"]


# (fixture, row ranges per K/V half of the zigzag mesh fetch): the row-range waves exist at ring degree > 2 only
# the third item: the pair exchange striped over the other ranks (comm/relay_exchange.py), a THIRD kind of traffic
# (world-group send/recv) beside the two communicators' -- where it applies (ulysses 2, at least two helpers)
# the fourth item: "barrier" = every collective is a host rendezvous of its whole group (one grouped RCCL call for all
# members); "pairwise" = no rendezvous: every message is matched between its two ranks only and the ranks drift apart on
# the host (random delays) the way real ranks do -- a schedule whose ranks post in different orders times out there
GRID_CASES = [(f, w, False, "barrier") for f in GRID for w in ((1, 2) if Golden(f).impl == "zigzag" and Golden(f).rd > 2 else (1,))] + \
             [(f, 1, True, "barrier") for f in GRID if Golden(f).ud == 2] + \
             [(f, 2 if Golden(f).rd > 2 else 1, relay, "pairwise") for f in GRID for relay in (False, True) if Golden(f).ud == 2]


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("path,pieces,relay,harness", GRID_CASES,
                         ids=lambda v: v.split("/")[-1][:-4] if isinstance(v, str) and "/" in v else
                         (v if isinstance(v, str) else "relay" if v is True else "direct" if v is False else f"pieces{v}"))
def test_pipelined_exchange_beside_a_ring_through_rccl(nccl_single, monkeypatch, path, pieces, relay, harness):
    """The two-communicator schedule (USP_PIPELINE_ULYSSES=1 beside a ring: BASELINE's 8-GPU grid ulysses 2 x ring 4,
    GQA, zigzag, forward + backward; and the 2 x 2 grids) with every exchange and every ring transfer going through real
    RCCL, stream-ordered only: head group i's ring attention starts behind ITS exchange, its output exchange runs
    behind group i+1's kernels, the zigzag fetch posts its waves (USP_ZZ_PIECES) beside them, the dK/dV accumulators
    travel from the compute streams.  Iterations must be bit-identical and equal to the reference's run."""
    from golden_util import grad_tol
    from virtual_grid import patch_dist, run_grid
    dev = torch.device("cuda:0")
    g = Golden(path)
    from virtual_grid import VirtualGridPairwise
    grid = _VirtualGrid(g.ud, g.rd, nccl_single) if harness == "barrier" else VirtualGridPairwise(g.ud, g.rd, nccl_single, jitter=(11, 0.002))
    AL = patch_dist(monkeypatch, grid)
    monkeypatch.setattr(AL, "_FILL_ITEMS", 1)                # tiny fixture: let the head groups form
    monkeypatch.setenv("USP_ZZ_PIECES", str(pieces))
    import yunchang_amd.comm.relay_exchange as RX
    monkeypatch.setitem(RX._OVERRIDE, "relay", relay)
    dtype = getattr(torch, g.dtype)
    ws = g.ws
    loc = [[torch.from_numpy(np.ascontiguousarray(g.shard(x, r))).to(dtype).to(dev) for x in (g.q, g.k, g.v, g.dout)]
           for r in range(ws)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(ws)]
    torch.cuda.synchronize()

    def rank_fn(r):
        torch.cuda.set_device(dev)
        q, k, v, do = loc[r]
        upg, rpg = grid.groups_of(r)
        ctx = _Ctx()
        with torch.cuda.stream(streams[r]):
            out = AL._AsyncUSPFunc.forward(ctx, q, k, v, None, g.causal, upg, rpg, g.impl, AL._MAX_GROUPS)
            grads = AL._AsyncUSPFunc.backward(ctx, do)[:3] if g.bwd else ()
        return (out,) + tuple(grads)

    names = ("out", "dq", "dk", "dv")
    first = None
    for it in range(8):
        res = run_grid(grid, ws, rank_fn)
        torch.cuda.synchronize()
        got = [[t.clone() for t in res[r]] for r in range(ws)]
        if first is None:
            first = got
            # both communicators carried traffic (relayed: the pair exchanges ride world-group send/recv instead)
            assert {k for k, _ in grid.calls} == ({"world", "ring"} if relay else {"ulysses", "ring"})
            for r in range(ws):
                for t, name in zip(got[r], names):
                    tol = TOL[g.dtype]["out"] if name == "out" else grad_tol(g.dtype, g.Hq // g.Hkv if name != "dq" else 1)
                    assert_close(t.float().cpu().numpy(), getattr(g, name)[r], *tol, f"{g.name} {name} rank {r}")
        else:
            for r in range(ws):
                for t, t0, name in zip(got[r], first[r], names):
                    assert torch.equal(t, t0), f"iteration {it}: {name} of rank {r} differs from iteration 0"


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("n_gpus,relay,harness", [(8, False, "barrier"), (4, False, "barrier"), (2, False, "barrier"),
                                                  (8, True, "barrier"), (8, False, "pairwise"), (4, False, "pairwise"),
                                                  (-4, False, "barrier"), (-2, False, "barrier"), (-2, False, "barrier-selfchunk"),
                                                  (-8, False, "barrier-selfchunk"), (-8, False, "pairwise-selfchunk")],
                         ids=["configs4_8gpu_u2r4_gqa_fwd_bwd", "configs3_4gpu_r4_fwd", "configs2_2gpu_u2_fwd",
                              "configs4_8gpu_relayed_pair_exchange", "configs4_8gpu_drifting_ranks", "configs3_4gpu_drifting_ranks",
                              "bench_4gpu_r4_64k_gqa_fwd_bwd", "bench_2gpu_u2_64k_gqa_fwd_bwd", "bench_2gpu_u2_64k_self_chunk_start",
                              "bench_8gpu_u2r4_64k_self_chunk_start_and_tails", "bench_8gpu_u2r4_64k_defaults_drifting_ranks"])
def test_baseline_configs_at_full_size_on_a_virtual_grid(nccl_single, monkeypatch, n_gpus, relay, harness):
    """BASELINE's multi-GPU configs AT THEIR OWN SIZE with the ranks as virtual ranks of one GPU -- configs[4]: 8 ranks,
    ulysses 2 x ring 4, zigzag, B1 S65536 H32/Hkv4 D128 bf16 causal, forward + backward; configs[3]: 4 ranks, ring 4
    zigzag, S32768 H16, forward; configs[2]: 2 ranks, ulysses 2, S16384 H16, forward -- through the layer's default
    schedule (packed q|k|v exchange pipelined over head groups, beside the ring where there is one; zigzag mesh fetch in
    row-range waves; K split of few-head launches; travelling dK/dV with the rounded, pending last hop), the real kernels
    at the real per-rank shapes, every transfer a real RCCL call.  The shards are put back together and checked against
    the single-launch result of the same tensors and against exact fp64 attention over the whole sequence on sampled
    rows (out, dQ) and sampled key columns (dK, dV summed over the GQA group and every later row) -- bench.sampled_parity,
    the check the single-GPU 64K entry of the bench line carries."""
    import importlib.util
    from oracle import usp_oracle as O
    from yunchang_amd import _C
    from virtual_grid import patch_dist, run_grid
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    # n_gpus > 0: BASELINE.json's config for that GPU count; n_gpus < 0: what `bench.py --gpus |n|` runs (round 5: the metric's
    # configuration -- configs[4]'s global tensors, fwd+bwd -- on the grid BASELINE names for the count)
    metric = n_gpus < 0
    n_gpus = abs(n_gpus)
    cfg = (b.WORKLOADS if metric else b.CONFIG_WORKLOADS)[n_gpus]
    dev = torch.device("cuda:0")
    ud, rd, ws, impl, bwd = cfg["ud"], cfg["rd"], n_gpus, cfg["impl"], cfg["bwd"]
    q, k, v, do = b.make_global(cfg, dev)

    def ext(t, r):          # the shard of rank r (comm/extract_local.py needs the real process grid: restated, and held
        if impl == "basic":                                 # against the oracle's layout right below)
            return t.chunk(ws, dim=1)[r]
        ch = t.chunk(2 * rd, dim=1)
        return torch.cat([ch[r // ud], ch[2 * rd - 1 - r // ud]], dim=1).chunk(ud, dim=1)[r % ud]
    rows = torch.arange(cfg["S"], device=dev).view(1, -1, 1, 1)
    owned = [ext(rows, r).reshape(-1) for r in range(ws)]                                        # global row ids per rank
    small = np.arange(64, dtype=np.float32).reshape(1, 64, 1, 1)
    for r in range(ws):
        assert np.array_equal(ext(torch.from_numpy(small), r).numpy(), O.EXTRACT[impl](small, r, ws, rd, ud))
    loc = [[ext(t, r).contiguous() for t in (q, k, v, do)] for r in range(ws)]
    from virtual_grid import VirtualGridPairwise
    # "pairwise": no host rendezvous of a group at its collectives, random host delays -- the ranks drift apart (virtual_grid.py)
    grid = _VirtualGrid(ud, rd, nccl_single) if harness.startswith("barrier") else VirtualGridPairwise(ud, rd, nccl_single, jitter=(5, 0.004))
    AL = patch_dist(monkeypatch, grid)
    split_calls, tail_calls = [], []
    if not harness.endswith("selfchunk"):        # rounds 3-5's schedule (still selectable): no self-chunk start, no tails
        monkeypatch.setitem(AL._COMM_OVERRIDE, "self_chunk", "0")
        monkeypatch.setitem(AL._COMM_OVERRIDE, "tails", "0")
    if harness.endswith("selfchunk"):            # the library's defaults since round 6: head groups start on the rank's own rows,
        import yunchang_amd.comm.all_to_all as A_    # the last group's output leaves in row pieces, its dq ahead of dk | dv
        real_pack = A_.pack_seq_rows
        monkeypatch.setattr(A_, "pack_seq_rows", lambda *a: (tail_calls.append(a[2:]), real_pack(*a))[1])
        real_f, real_b = AL._split_first_forward, AL._split_first_backward
        monkeypatch.setattr(AL, "_split_first_forward", lambda *a: (split_calls.append("f"), real_f(*a))[1])
        monkeypatch.setattr(AL, "_split_first_backward", lambda *a: (split_calls.append("b"), real_b(*a))[1])
        import yunchang_amd.ring.zigzag_ring_flash_attn as ZZ           # beside a ring: step 0 of the ring schedule is split
        real_zf, real_zb = ZZ.zigzag_fwd_step0_own, ZZ.zigzag_bwd_step0_split
        monkeypatch.setattr(ZZ, "zigzag_fwd_step0_own", lambda *a: (split_calls.append("f"), real_zf(*a))[1])
        monkeypatch.setattr(ZZ, "zigzag_bwd_step0_split", lambda *a: (split_calls.append("b"), real_zb(*a))[1])
    import yunchang_amd.comm.relay_exchange as RX
    monkeypatch.setitem(RX._OVERRIDE, "relay", relay)         # (8 ranks: every pair exchange striped over the 6 other ranks)
    streams = [torch.cuda.Stream(device=dev) for _ in range(ws)]
    torch.cuda.synchronize()

    def rank_fn(r):
        torch.cuda.set_device(dev)
        lq, lk, lv, ldo = loc[r]
        upg, rpg = grid.groups_of(r)
        ctx = _Ctx()
        ctx.needs_input_grad = (bwd, bwd, bwd)                # forward-only configs: head groups sized for the K split
        with torch.cuda.stream(streams[r]):
            if ud == 1:       # no exchange: the layer hands q, k, v straight to the ring function (hybrid/attn_layer.py)
                import yunchang_amd.ring.zigzag_ring_flash_attn as Z
                o, l = Z.zigzag_ring_flash_attn_forward(rpg, lq, lk, lv, cfg["D"] ** -0.5)[:2]
                if not bwd:
                    return (o,), 1
                return (o,) + tuple(Z.zigzag_ring_flash_attn_backward(rpg, ldo, lq, lk, lv, o, l, cfg["D"] ** -0.5)), 1
            out = AL._AsyncUSPFunc.forward(ctx, lq, lk, lv, None, True, upg, rpg, impl, AL._MAX_GROUPS)
            grads = AL._AsyncUSPFunc.backward(ctx, ldo)[:3] if bwd else ()
        return (out,) + tuple(grads), ctx.meta[6]

    res = run_grid(grid, ws, rank_fn)
    torch.cuda.synchronize()
    if harness.endswith("selfchunk"):
        assert sorted(split_calls) == ["b"] * ws + ["f"] * ws, split_calls            # every rank took the split path, both passes
        # ... and the last group's output left in 4 row pieces on every rank (beside the ring: pieces of the final launch; at ring
        # degree 1: the last group's causal block issued in the layer as row-range launches)
        assert len(tail_calls) == 4 * ws and len(set(tail_calls)) == 4, tail_calls
    if not metric:
        assert {n for _, n in res} == {{8: 2, 4: 1, 2: 4}[n_gpus]}        # head groups per rank: the default pipeline
    kinds = {"ulysses", "ring"} - ({"ulysses"} if ud == 1 else set()) - ({"ring"} if rd == 1 else set())
    assert {kind for kind, _ in grid.calls} == ((kinds - {"ulysses"}) | {"world"} if relay else kinds)
    glob = [torch.empty_like(t) for t in ((q, q, k, v) if bwd else (q,))]      # out [, dq, dk, dv]
    for r in range(ws):
        for g, shard in zip(glob, res[r][0]):
            g[:, owned[r]] = shard
    # ONE single-GPU forward of the same tensors: its rows against the grid's, and its LSE as the key columns' softmax
    # normaliser (sampled_parity checks that LSE against the exact one on its sampled rows)
    lse = torch.empty((1, cfg["Hq"], cfg["S"]), dtype=torch.float32, device=dev)
    one = torch.empty_like(q)
    _C.flash_fwd(q, k, v, cfg["D"] ** -0.5, True, lse, one)
    d = (glob[0].float() - one.float()).abs()
    assert bool((d <= 2e-2 + 2e-2 * one.float().abs()).all()), float(d.max())
    if not bwd:
        glob += [torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)]
    err = b.sampled_parity(dict(q=q, k=k, v=v, do=do, out=glob[0], lse=lse, dq=glob[1], dk=glob[2], dv=glob[3]))["max_abs_err"]
    if not bwd:
        err = {n: err[n] for n in ("out", "lse")}
    print(f"{cfg['name']}: virtual grid at full size, max abs errors vs fp64 samples:", err)      # (pytest -rP shows it)
    assert err["out"] < 2e-2 and err["lse"] < 2e-3, err
    if bwd:
        g = cfg["Hq"] // cfg["Hkv"]
        assert err["dq"] < 5e-2 and err["dk"] < 5e-2 * g ** 0.5 and err["dv"] < 5e-2 * g ** 0.5, err
