"""N > 1 path on CPU: the package's own LongContextAttention / SeqAllToAll4D / ring schedules /
KVRelay / RingComm run under torch.distributed + gloo with the CPU oracle plugged in as the block
kernel (tests/oracle_backend.py), and are compared with
  (a) the golden fixtures produced by the REFERENCE on the same grid (tests/golden), and
  (b) plain attention on the unsharded tensors (size-independent property).
"""
import numpy as np
import pytest
import torch

from dist_util import run_distributed
from golden_util import Golden, TOL, assert_close, golden_files


def _usp_worker(rank, ws, path, use_autograd):
    import torch.distributed as dist
    import yunchang_amd as Y
    from yunchang_amd.kernels import set_block_backend
    from oracle_backend import OracleBlockBackend

    g = Golden(path)
    be = OracleBlockBackend()
    set_block_backend(be)
    import yunchang_amd.ring.utils as U
    direct_calls, real_direct = [], U.return_dkdv_direct
    U.return_dkdv_direct = lambda *a, **k: (direct_calls.append(1), real_direct(*a, **k))[1]
    dtype = getattr(torch, g.dtype)
    Y.set_seq_parallel_pg(g.ud, g.rd, rank, ws)
    ext = Y.EXTRACT_FUNC_DICT[g.impl]
    glob = [torch.from_numpy(t).to(dtype) for t in (g.q, g.k, g.v, g.dout)]
    lq, lk, lv, ldo = (ext(t, rank, world_size=ws, rd=g.rd, ud=g.ud).detach().clone() for t in glob)
    if g.bwd:
        for t in (lq, lk, lv):
            t.requires_grad_(True)
    kw = dict(dropout_p=0, causal=g.causal, window_size=(-1, -1), softcap=0.0, alibi_slopes=None,
              deterministic=False, return_attn_probs=True)
    qkv = None
    if g.layer == "hybrid":
        attn = Y.LongContextAttention(ring_impl_type=g.impl, attn_type=Y.AttnType.TORCH_EFFICIENT)
        out = attn(lq, lk, lv, **kw)
    elif g.layer == "ulysses":
        attn = Y.UlyssesAttention(Y.PROCESS_GROUP.ULYSSES_PG, attn_type=Y.AttnType.HIP)
        out = attn(lq, lk, lv, **kw)
    else:
        qkv = torch.stack([lq.detach(), lk.detach(), lv.detach()], dim=2).requires_grad_(g.bwd)
        attn = Y.LongContextAttentionQKVPacked(ring_impl_type=g.impl, attn_type=Y.AttnType.HIP)
        out = attn(qkv, **kw)
    res = {"out": out.detach().float().numpy(), "calls": list(be.calls), "direct": direct_calls}
    if g.bwd:
        out.backward(ldo)
        if qkv is not None:
            gq, gk, gv = qkv.grad[:, :, 0], qkv.grad[:, :, 1], qkv.grad[:, :, 2]
        else:
            gq, gk, gv = lq.grad, lk.grad, lv.grad
        res.update(dq=gq.float().numpy(), dk=gk.float().numpy(), dv=gv.float().numpy())
    return res


MULTI = [f for f in golden_files() if "_w1" not in f]


@pytest.mark.parametrize("path", MULTI, ids=lambda p: p.split("/")[-1][:-4])
def test_usp_on_gloo_matches_reference_golden(path):
    g = Golden(path)
    res = run_distributed(_usp_worker, g.ws, path, True)
    from oracle import usp_oracle as O
    full, _ = O.attention_ref(g.q, g.k, g.v, causal=g.causal)
    atol, rtol = TOL[g.dtype]["out"]
    for r in range(g.ws):
        assert_close(res[r]["out"], g.out[r], atol, rtol, f"{g.name} out rank {r} vs reference run")
        # vs exact attention on the unsharded tensors (only output rounding separates them)
        assert_close(res[r]["out"], g.shard(full, r), atol / 2, rtol / 2, f"{g.name} out rank {r} vs truth")
    if g.bwd:
        atol, rtol = TOL[g.dtype]["grad"]
        for r in range(g.ws):
            for key in ("dq", "dk", "dv"):
                assert_close(res[r][key], getattr(g, key)[r], atol, rtol, f"{g.name} {key} rank {r}")


@pytest.mark.parametrize("pieces", [None, 2, 3])
def test_zigzag_schedule_block_shapes(pieces, monkeypatch):
    """Every ring step of the zigzag schedule costs the same 2c^2 score entries (the load-balance property the layout
    exists for), whatever the number of row ranges the mesh fetch cuts a K/V half into (ring/utils.py:ZigzagKVFetch;
    default 1 at this size, 3 = uneven cuts), launches come wave by wave, and every row is emitted exactly once --
    by the LAST launch that touches it."""
    if pieces:
        monkeypatch.setenv("USP_ZZ_PIECES", str(pieces))
    W = pieces or 1
    path = [f for f in MULTI if "c4_w4_u1r4" in f][0]
    g = Golden(path)
    res = run_distributed(_usp_worker, g.ws, path, False)
    c = g.S // (2 * g.rd)
    full, _ = __import__("oracle.usp_oracle", fromlist=["x"]).attention_ref(g.q, g.k, g.v, causal=True)
    for r in range(g.ws):
        assert_close(res[r]["out"], g.shard(full, r), *[t / 2 for t in TOL[g.dtype]["out"]], f"pieces {pieces} rank {r}")
        fwd = [x for x in res[r]["calls"] if x[0] == "fwd"]
        # step 0, then per wave (batch 1: the pieces of a wave's source ranks lie one behind the other and the steps
        # that share a query range are ONE launch): every q row x steps 1..r (front waves), q[c:] x steps r+1..P-1
        assert len(fwd) == 1 + W * (r >= 1) + 2 * W * (r < g.rd - 1)
        _, qs, ks, causal, merge_in, fb, fe = fwd[0]
        assert causal and not merge_in and qs[1] == ks[1] == 2 * c
        assert sum(x[1][1] * x[2][1] for x in fwd[1:]) == (g.rd - 1) * 2 * c * c
        assert all(not x[3] and x[4] for x in fwd[1:])                      # later launches: full blocks, merged in
        key_rows = [x[2][1] for x in fwd[1:]]                               # wave-major: piece sizes never interleave
        cuts = [(i + 1) * c // W - i * c // W for i in range(W)]
        per_wave = lambda n, back: ([n * r] if (r >= 1 and not back) else []) + ([n * (g.rd - 1 - r)] if r < g.rd - 1 else [])
        want = [x for n in cuts for x in per_wave(n, False)] + [x for n in cuts for x in per_wave(n, True)]
        assert key_rows == want
        done = np.zeros(2 * c, bool)                                        # rows already emitted in 16 bits
        for _, qs, ks, causal, merge_in, fb, fe in fwd:
            off = 2 * c - qs[1]                                             # q[c:] launches address rows c..2c
            assert not done[off:].any(), "a launch touches rows that were already final"
            done[off + fb:off + fe] = True
        assert done.all()


def _a2a_worker(rank, ws):
    import torch.distributed as dist
    import yunchang_amd as Y
    from yunchang_amd.comm.all_to_all import SeqAllToAll4D, all_to_all_4D
    Y.set_seq_parallel_pg(ws, 1, rank, ws)
    pg = Y.PROCESS_GROUP.ULYSSES_PG
    B, Sl, H, D = 2, 6, 4 * ws, 8
    torch.manual_seed(rank)
    x = torch.randn(B, Sl, H, D)
    allx = [torch.empty_like(x) for _ in range(ws)]
    dist.all_gather(allx, x)
    y = all_to_all_4D(x, 2, 1, group=pg)
    hp = H // ws
    want = torch.cat([t[:, :, rank * hp:(rank + 1) * hp] for t in allx], dim=1)
    assert y.is_contiguous() and torch.equal(y, want)
    z = all_to_all_4D(y, 1, 2, group=pg)
    assert torch.equal(z, x)                      # round trip
    # autograd: gradient of the exchange is the inverse exchange
    xr = x.clone().requires_grad_(True)
    yy = SeqAllToAll4D.apply(pg, xr, 2, 1, False)
    w = torch.randn(yy.shape)
    (yy * w).sum().backward()
    wz = all_to_all_4D(w, 1, 2, group=pg)
    assert torch.allclose(xr.grad, wz)
    return True


@pytest.mark.parametrize("ws", [2, 4])
def test_all_to_all_round_trip(ws):
    assert all(run_distributed(_a2a_worker, ws))


def _async_worker(rank, ws, ud, rd, impl, Hq, Hkv):
    import os
    import yunchang_amd as Y
    import yunchang_amd.hybrid.async_attn_layer as AL
    from yunchang_amd.kernels import set_block_backend
    from oracle_backend import OracleBlockBackend
    from oracle import usp_oracle as O
    set_block_backend(OracleBlockBackend())
    Y.set_seq_parallel_pg(ud, rd, rank, ws)
    AL._FILL_ITEMS = 1              # tiny problem: let the head-group pipeline form anyway
    torch.manual_seed(0)
    B, S, D = 2, 32 * ws, 32
    q, k, v, do = (torch.randn(B, S, h, D).to(torch.bfloat16) for h in (Hq, Hkv, Hkv, Hq))
    ext = Y.EXTRACT_FUNC_DICT[impl]
    # truth: exact attention and its gradients on the unsharded tensors (fp64 oracle), sharded like the inputs
    qn, kn, vn, don = (t.float().numpy().astype(np.float64) for t in (q, k, v, do))
    ro, rl = O.attention_ref(qn, kn, vn, causal=True)
    truth = [ext(torch.from_numpy(np.ascontiguousarray(t)), rank, world_size=ws, rd=rd, ud=ud).float()
             for t in (ro,) + tuple(O.block_bwd(don, qn, kn, vn, ro, rl, None, True))]
    res = []
    # (the claim is about REGROUPING the exchange: the self-chunk start and the row-chunked tails -- defaults since round 6 --
    # change the launches and with them the fp32 summation order; they have tests of their own below)
    AL._COMM_OVERRIDE.update(self_chunk="0", tails="0")
    # the pipelined packed exchange (default), one packed exchange, the reference's three exchanges, the async layer
    # (+ the layer's default and USP_SAFE_COMM=1, which beside a ring are the one-exchange form)
    for cls, env in ((Y.LongContextAttention, {"USP_PIPELINE_ULYSSES": "1"}), (Y.LongContextAttention, {"USP_PIPELINE_ULYSSES": "0"}),
                     (Y.LongContextAttention, {"USP_PACK_QKV": "0"}), (Y.AsyncLongContextAttention, {}),
                     (Y.LongContextAttention, {}), (Y.LongContextAttention, {"USP_SAFE_COMM": "1", "USP_PIPELINE_ULYSSES": "1"})):
        lq, lk, lv, ldo = (ext(t, rank, world_size=ws, rd=rd, ud=ud).detach().clone() for t in (q, k, v, do))
        for t in (lq, lk, lv):
            t.requires_grad_(True)
        os.environ.update(env)
        try:
            out = cls(ring_impl_type=impl)(lq, lk, lv, causal=True)
            out.backward(ldo)
        finally:
            for key in env:
                del os.environ[key]
        res.append([t.detach().float() for t in (out, lq.grad, lk.grad, lv.grad)])
    AL._COMM_OVERRIDE.clear()
    # same maths per head, only the exchange is regrouped: the four must be identical ...
    same = all(torch.equal(a, b) for other in res[1:] for a, b in zip(res[0], other))
    # ... and right (bf16 tolerances of golden_util.TOL: out 2e-2, grads 5e-2)
    right = all(torch.allclose(a, t, atol=tol, rtol=tol) for a, t, tol in zip(res[0], truth, (2e-2, 5e-2, 5e-2, 5e-2)))
    return same and right


def _self_chunk_worker(rank, ws, impl, Hq, Hkv, B, S, rd=1, env=None):
    """USP_SELF_CHUNK=1 (hybrid/async_attn_layer.py:self_chunk_mode): at ulysses degree 2 the first head group starts on the
    rows the rank already holds -- on the 2-GPU grid (ring degree 1) two or three launches joined by the fused LSE merge /
    fp32 dK, dV accumulation instead of one causal block; beside a zigzag ring (rd > 1: the 8-GPU grid is 2 x 4) the same
    split of STEP 0 of the ring schedule, the ring's own K/V transfers posted behind the exchange.  Against the unsplit
    schedule (same sums in another order: tight) and against exact attention and its gradients; the launches the test
    backend sees must START before the exchange is waited for."""
    import yunchang_amd as Y
    import yunchang_amd.hybrid.async_attn_layer as AL
    import yunchang_amd.ring.zigzag_ring_flash_attn as ZZ
    from yunchang_amd.kernels import set_block_backend
    from oracle_backend import OracleBlockBackend
    from oracle import usp_oracle as O
    import os
    for kv in (env or ()):
        os.environ.__setitem__(*kv.split("="))
    set_block_backend(OracleBlockBackend())
    Y.set_seq_parallel_pg(2, rd, rank, ws)
    AL._FILL_ITEMS = 1
    torch.manual_seed(1)
    D = 32
    q, k, v, do = (torch.randn(B, S, h, D).to(torch.bfloat16) for h in (Hq, Hkv, Hkv, Hq))
    ext = Y.EXTRACT_FUNC_DICT[impl]
    qn, kn, vn, don = (t.float().numpy().astype(np.float64) for t in (q, k, v, do))
    ro, rl = O.attention_ref(qn, kn, vn, causal=True)
    truth = [ext(torch.from_numpy(np.ascontiguousarray(t)), rank, world_size=ws, rd=rd, ud=2).float()
             for t in (ro,) + tuple(O.block_bwd(don, qn, kn, vn, ro, rl, None, True))]
    res, order = [], []
    real_wait = AL._Lane.wait
    real = (AL._split_first_forward, AL._split_first_backward, ZZ.zigzag_fwd_step0_own, ZZ.zigzag_bwd_step0_split)
    AL._Lane.wait = lambda self, ev: (order.append("wait"), real_wait(self, ev))[1]

    def spy(name, fn):
        def f(*a, **kw):
            order.append(name)
            return fn(*a, **kw)
        return f
    AL._split_first_forward, AL._split_first_backward = spy("split-forward", real[0]), spy("split-backward", real[1])
    ZZ.zigzag_fwd_step0_own, ZZ.zigzag_bwd_step0_split = spy("split-forward", real[2]), spy("split-backward", real[3])
    try:
        for on in (False, True):
            AL._COMM_OVERRIDE["self_chunk"] = "1" if on else "0"
            lq, lk, lv, ldo = (ext(t, rank, world_size=ws, rd=rd, ud=2).detach().clone() for t in (q, k, v, do))
            for t in (lq, lk, lv):
                t.requires_grad_(True)
            order.clear()
            out = Y.LongContextAttention(ring_impl_type=impl)(lq, lk, lv, causal=True)
            fwd_order = list(order)
            order.clear()
            out.backward(ldo)
            bwd_order = list(order)
            res.append([t.detach().float() for t in (out, lq.grad, lk.grad, lv.grad)])
            if on:      # the split step is entered BEFORE the first wait for an exchange, forward and backward
                assert fwd_order and fwd_order[0] == "split-forward" and "wait" in fwd_order[1:], fwd_order
                assert bwd_order and bwd_order[0] == "split-backward" and "wait" in bwd_order[1:], bwd_order
            else:
                assert "split-forward" not in fwd_order and "split-backward" not in bwd_order
    finally:
        AL._COMM_OVERRIDE.pop("self_chunk", None)
        AL._Lane.wait = real_wait
        AL._split_first_forward, AL._split_first_backward, ZZ.zigzag_fwd_step0_own, ZZ.zigzag_bwd_step0_split = real
    close = all(torch.allclose(a, b, atol=tol, rtol=tol) for a, b, tol in zip(res[0], res[1], (8e-3, 3e-2, 3e-2, 3e-2)))
    right = all(torch.allclose(a, t, atol=tol, rtol=tol) for a, t, tol in zip(res[1], truth, (2e-2, 5e-2, 5e-2, 5e-2)))
    return close and right


@pytest.mark.parametrize("impl,Hq,Hkv,B,S", [("basic", 4, 4, 1, 64), ("basic", 8, 2, 2, 96), ("zigzag", 4, 2, 1, 128)])
def test_self_chunk_start_on_the_two_gpu_grid(impl, Hq, Hkv, B, S):
    assert all(run_distributed(_self_chunk_worker, 2, impl, Hq, Hkv, B, S))


@pytest.mark.parametrize("rd,Hq,Hkv,B,S,env", [(2, 4, 2, 1, 128, None),                               # chain relay (ring 2)
                                               (4, 8, 2, 2, 256, None),                               # the 8-GPU grid: mesh fetch in waves
                                               (4, 4, 4, 1, 128, ("USP_KV_RELAY=chain", "USP_DKDV_RETURN=direct")),
                                               (4, 4, 2, 1, 128, ("USP_PIPELINE_ULYSSES=0",))])        # one packed exchange
def test_self_chunk_start_beside_a_zigzag_ring(rd, Hq, Hkv, B, S, env):
    """The same start at ring degree > 1 (ulysses 2 x ring 2 / 4, zigzag): step 0 of the ring schedule is split, every K/V
    transport (chain relay, mesh fetch) and both dK/dV return forms behind it."""
    assert all(run_distributed(_self_chunk_worker, 2 * rd, "zigzag", Hq, Hkv, B, S, rd, env))


def _tails_worker(rank, ws, rd, Hq, Hkv, B, S, n, env, sc="1", impl="zigzag"):
    """Row-chunked tails (hybrid/async_attn_layer.py:tails_mode; round 6, default beside a zigzag ring at ulysses degree 2): the
    LAST head group's last forward launch runs in n row pieces, each followed by an exchange of its rows; the last ring step
    of its backward issues dQ first and dq travels ahead of dk | dv; beside that, every group's owned chunk is launched in
    front of the first wait.  Against the schedule without tails and self-chunk (same sums, another order: tight), against
    exact attention and its gradients; the exchanges must be posted INSIDE the ring pass, in the same number on every rank."""
    import os
    import yunchang_amd as Y
    import yunchang_amd.hybrid.async_attn_layer as AL
    import yunchang_amd.comm.all_to_all as A
    import yunchang_amd.ring.zigzag_ring_flash_attn as ZZ
    from yunchang_amd.kernels import set_block_backend
    from oracle_backend import OracleBlockBackend
    from oracle import usp_oracle as O
    for kv in (env or ()):
        os.environ.__setitem__(*kv.split("="))
    be = OracleBlockBackend()
    set_block_backend(be)
    Y.set_seq_parallel_pg(2, rd, rank, ws)
    AL._FILL_ITEMS = 1
    torch.manual_seed(3)
    D = 32
    q, k, v, do = (torch.randn(B, S, h, D).to(torch.bfloat16) for h in (Hq, Hkv, Hkv, Hq))
    ext = Y.EXTRACT_FUNC_DICT[impl]
    qn, kn, vn, don = (t.float().numpy().astype(np.float64) for t in (q, k, v, do))
    ro, rl = O.attention_ref(qn, kn, vn, causal=True)
    truth = [ext(torch.from_numpy(np.ascontiguousarray(t)), rank, world_size=ws, rd=rd, ud=2).float()
             for t in (ro,) + tuple(O.block_bwd(don, qn, kn, vn, ro, rl, None, True))]
    res, log = [], []
    real_x, real_wait = A._exchange, AL._Lane.wait

    def spy_x(send, group, use_sync):
        log.append(("exchange", tuple(send.shape), len(be.calls)))
        return real_x(send, group, use_sync)
    A._exchange = spy_x
    AL._Lane.wait = lambda self, ev: (log.append(("wait", len(be.calls))), real_wait(self, ev))[1]
    counts = []
    try:
        for on in (False, True):
            AL._COMM_OVERRIDE.update(self_chunk=sc if on else "0", tails=str(n) if on else "0")
            lq, lk, lv, ldo = (ext(t, rank, world_size=ws, rd=rd, ud=2).detach().clone() for t in (q, k, v, do))
            for t in (lq, lk, lv):
                t.requires_grad_(True)
            log.clear(); be.calls.clear()
            out = Y.LongContextAttention(ring_impl_type=impl)(lq, lk, lv, causal=True)
            n_fwd_calls = len(be.calls)
            fwd_log = list(log)
            log.clear()
            out.backward(ldo)
            bwd_log = list(log)
            res.append([t.detach().float() for t in (out, lq.grad, lk.grad, lv.grad)])
            fx = [e for e in fwd_log if e[0] == "exchange"]
            bx = [e for e in bwd_log if e[0] == "exchange"]
            counts.append((len(fx), len(bx)))
            if on:
                ng = Hkv // 2
                c = S // (2 * rd)
                pieces = min(n, c)
                # forward: ng input exchanges, ng - 1 whole output exchanges, `pieces` row pieces of the last group's output,
                # all but the last piece posted while kernel calls were still to come
                assert len(fx) == 2 * ng - 1 + pieces, (fx, ng, pieces)
                assert sum(e[1][1] for e in fx[-pieces:]) == c, fx          # the pieces' rows make up a chunk
                assert all(e[2] < n_fwd_calls for e in fx[-pieces:-1]) and fx[-1][2] == n_fwd_calls, (fx, n_fwd_calls)
                # the first group's owned chunk (USP_SELF_CHUNK=all: every group's) is launched in front of the first wait
                first_wait = next(e for e in fwd_log if e[0] == "wait")
                assert first_wait[1] >= (ng if sc == "all" else 1), (fwd_log[:6], ng)
                # backward: ng input exchanges, ng - 1 packed gradient exchanges, then dq alone and dk | dv alone
                assert len(bx) == 2 * ng + 1, bx
                assert bx[-2][1][3] == Hq // Hkv and bx[-1][1][3] == 2, bx
                only = [c_[4] for c_ in be.calls if c_[0] == "bwd" and len(c_) == 5]
                # every ring step is issued dK/dV | hop | dQ (travel_dkdv: split_block); the LAST step of the last group dQ first
                assert only[-2:] == ["dq", "dkdv"] and all(only[i:i + 2] == ["dkdv", "dq"] for i in range(0, len(only) - 2, 2)), only
    finally:
        AL._COMM_OVERRIDE.clear()
        A._exchange, AL._Lane.wait = real_x, real_wait
    close = all(torch.allclose(a, b, atol=tol, rtol=tol) for a, b, tol in zip(res[0], res[1], (8e-3, 3e-2, 3e-2, 3e-2)))
    right = all(torch.allclose(a, t, atol=tol, rtol=tol) for a, t, tol in zip(res[1], truth, (2e-2, 5e-2, 5e-2, 5e-2)))
    return close and right, counts[1]


@pytest.mark.parametrize("rd,Hq,Hkv,B,S,n,env,sc", [(2, 8, 4, 1, 128, 2, None, "1"),                      # chain relay (ring 2), two head groups
                                                    (4, 8, 4, 1, 256, 4, None, "1"),                      # the 8-GPU grid: mesh fetch, grouped launches
                                                    (4, 8, 4, 1, 256, 4, None, "all"),                    # ... every group's owned chunk first
                                                    (4, 8, 2, 2, 256, 3, None, "1"),                      # batch 2 (ungrouped pieces), ONE head group, uneven pieces
                                                    (4, 4, 4, 1, 128, 2, ("USP_KV_RELAY=chain", "USP_DKDV_RETURN=direct"), "all"),
                                                    (1, 8, 4, 1, 128, 4, None, "1"),                      # the 2-GPU grid (ring degree 1): the block in the layer
                                                    (1, 8, 4, 2, 96, 3, None, "basic")])                  # ... contiguous layout, batch 2, uneven pieces
def test_row_chunked_tails_beside_a_zigzag_ring(rd, Hq, Hkv, B, S, n, env, sc):
    impl = "basic" if sc == "basic" else "zigzag"
    res = run_distributed(_tails_worker, 2 * rd, rd, Hq, Hkv, B, S, n, env, "1" if sc == "basic" else sc, impl)
    assert all(r[0] for r in res)
    assert len({r[1] for r in res}) == 1, "every rank posts the same number of exchanges"


@pytest.mark.parametrize("ws,ud,rd,impl,Hq,Hkv", [(4, 2, 2, "zigzag", 8, 4), (2, 2, 1, "basic", 4, 4),
                                                  (4, 4, 1, "basic", 16, 8), (4, 2, 2, "basic", 4, 4),
                                                  (4, 1, 4, "zigzag", 4, 2),      # ring 4: two-wave mesh fetch, B = 2
                                                  (8, 4, 2, "zigzag", 8, 4),      # ulysses 4 beside a ring of 2
                                                  (8, 8, 1, "basic", 8, 8),       # ulysses 8
                                                  (8, 1, 8, "zigzag", 4, 2),      # ring 8: two-wave mesh fetch over 7 peers
                                                  (8, 1, 8, "basic", 2, 2),       # ring 8, direct K/V fetch
                                                  (8, 2, 4, "strip", 4, 2)])      # stripe layout beside ulysses 2
def test_exchange_variants_agree_and_match_exact_attention(ws, ud, rd, impl, Hq, Hkv):
    """LongContextAttention with the pipelined packed exchange (default), with one packed exchange, with the
    reference's three separate exchanges, and AsyncLongContextAttention: bit-identical
    results, forward and backward, at batch 2 (seq-major views with real batch strides reach the ring at
    ulysses x ring = 2 x 2), with GQA, and equal to exact attention on the unsharded tensors."""
    assert all(run_distributed(_async_worker, ws, ud, rd, impl, Hq, Hkv))


def _other_layers_worker(rank, ws, layer, ud, rd, impl, Hq, Hkv, causal):
    """UlyssesAttention and LongContextAttentionQKVPacked at world size 8, where no
    reference run exists: exact attention on the unsharded tensors (fp64 oracle) is the truth."""
    import yunchang_amd as Y
    from yunchang_amd.kernels import set_block_backend
    from oracle_backend import OracleBlockBackend
    from oracle import usp_oracle as O
    set_block_backend(OracleBlockBackend())
    Y.set_seq_parallel_pg(ud, rd, rank, ws)
    torch.manual_seed(2)
    B, S, D = 2, 16 * ws, 32
    q, k, v, do = (torch.randn(B, S, h, D).to(torch.bfloat16) for h in (Hq, Hkv, Hkv, Hq))
    ext = Y.EXTRACT_FUNC_DICT[impl]
    qn, kn, vn, don = (t.float().numpy().astype(np.float64) for t in (q, k, v, do))
    ro, rl = O.attention_ref(qn, kn, vn, causal=causal)
    truth = [ext(torch.from_numpy(np.ascontiguousarray(t)), rank, world_size=ws, rd=rd, ud=ud).float()
             for t in (ro,) + tuple(O.block_bwd(don, qn, kn, vn, ro, rl, None, causal))]
    lq, lk, lv, ldo = (ext(t, rank, world_size=ws, rd=rd, ud=ud).detach().clone() for t in (q, k, v, do))
    if layer == "ulysses":
        for t in (lq, lk, lv):
            t.requires_grad_(True)
        out = Y.UlyssesAttention(Y.PROCESS_GROUP.ULYSSES_PG, attn_type=Y.AttnType.HIP)(lq, lk, lv, causal=causal)
        out.backward(ldo)
        got = [out, lq.grad, lk.grad, lv.grad]
    else:
        qkv = torch.stack([lq, lk, lv], dim=2).requires_grad_(True)
        out = Y.LongContextAttentionQKVPacked(ring_impl_type=impl, attn_type=Y.AttnType.HIP)(qkv, causal=causal)
        out.backward(ldo)
        got = [out, qkv.grad[:, :, 0], qkv.grad[:, :, 1], qkv.grad[:, :, 2]]
    got = [t.detach().float() for t in got]
    return all(torch.allclose(a, t, atol=tol, rtol=tol) for a, t, tol in zip(got, truth, (2e-2, 5e-2, 5e-2, 5e-2)))


@pytest.mark.parametrize("ws,layer,ud,rd,impl,Hq,Hkv,causal", [(8, "ulysses", 8, 1, "basic", 16, 8, True),
                                                              (4, "ulysses", 4, 1, "basic", 4, 4, False),
                                                              (8, "qkvpacked", 4, 2, "zigzag", 8, 8, True),
                                                              (8, "qkvpacked", 2, 4, "basic", 4, 4, False)])
def test_ulysses_and_qkvpacked_layers_match_exact_attention(ws, layer, ud, rd, impl, Hq, Hkv, causal):
    assert all(run_distributed(_other_layers_worker, ws, layer, ud, rd, impl, Hq, Hkv, causal))


# ---- packed variable-length ring schedules -------------------------------------
def _varlen_worker(rank, ws, path, packed_qkv):
    import torch.distributed as dist
    import yunchang_amd as Y
    from yunchang_amd.kernels import set_block_backend
    from oracle_backend import OracleBlockBackend
    from golden_util import VarlenGolden

    g = VarlenGolden(path)
    set_block_backend(OracleBlockBackend())
    dtype = getattr(torch, g.dtype)
    layout = "zigzag" if g.impl == "zigzag" else "basic"
    lq, lk, lv, ldo = (Y.extract_local_varlen(torch.from_numpy(t).to(dtype), g.cu, rank, ws, layout)
                       for t in (g.q, g.k, g.v, g.dout))
    cu_local = torch.tensor(g.cu_local, dtype=torch.int32)
    kw = dict(dropout_p=0.0, causal=True, window_size=(-1, -1), alibi_slopes=None, deterministic=False,
              return_attn_probs=True, group=dist.group.WORLD)
    if packed_qkv:       # equal head counts only: exercises the strided-view (qkv[:, i]) path
        qkv = torch.stack([lq, lk, lv], dim=1).requires_grad_(True)
        fn = (Y.zigzag_ring_flash_attn_varlen_qkvpacked_func if g.impl == "zigzag"
              else Y.ring_flash_attn_varlen_qkvpacked_func)
        out, lse, _ = fn(qkv, cu_local, g.max_local, **kw)
        out.backward(ldo)
        gq, gk, gv = qkv.grad[:, 0], qkv.grad[:, 1], qkv.grad[:, 2]
    else:
        for t in (lq, lk, lv):
            t.requires_grad_(True)
        fn = Y.zigzag_ring_flash_attn_varlen_func if g.impl == "zigzag" else Y.ring_flash_attn_varlen_func
        out, lse, _ = fn(lq, lk, lv, cu_local, g.max_local, **kw)
        out.backward(ldo)
        gq, gk, gv = lq.grad, lk.grad, lv.grad
    assert lse.shape == (len(g.lens), g.Hq, g.max_local)          # the reference's padded layout
    return dict(out=out.detach().float().numpy(), lse=Y.flatten_lse(lse.detach(), cu_local).numpy(),
                dq=gq.float().numpy(), dk=gk.float().numpy(), dv=gv.float().numpy())


from golden_util import varlen_golden_files  # noqa: E402


@pytest.mark.parametrize("path", varlen_golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_varlen_ring_on_gloo_matches_reference_golden(path):
    from golden_util import VarlenGolden
    g = VarlenGolden(path)
    res = run_distributed(_varlen_worker, g.ws, path, g.Hq == g.Hkv and "oneseq" in g.name)
    atol, rtol = TOL[g.dtype]["out"]
    gt, gr = TOL[g.dtype]["grad"]
    for r in range(g.ws):
        assert_close(res[r]["out"], g.out[r], atol, rtol, f"{g.name} out rank {r}")
        assert_close(res[r]["lse"], g.lse[r], atol, rtol, f"{g.name} lse rank {r}")
        for key in ("dq", "dk", "dv"):
            assert_close(res[r][key], getattr(g, key)[r], gt, gr, f"{g.name} {key} rank {r}")


def _varlen_truth_worker(rank, ws, impl, lens, Hq, Hkv):
    """No reference run has ring degree 8: exact causal attention per sequence (fp64 oracle) is the truth here."""
    import torch.distributed as dist
    import yunchang_amd as Y
    from yunchang_amd.kernels import set_block_backend
    from oracle_backend import OracleBlockBackend
    from oracle import usp_oracle as O
    set_block_backend(OracleBlockBackend())
    torch.manual_seed(1)
    D, T = 32, sum(lens)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    q, k, v, do = (torch.randn(T, h, D).to(torch.bfloat16) for h in (Hq, Hkv, Hkv, Hq))
    layout = "zigzag" if impl == "zigzag" else "basic"
    truth = [np.zeros((T, h, D)) for h in (Hq, Hq, Hkv, Hkv)]                 # out, dq, dk, dv
    for a, b in zip(cu[:-1], cu[1:]):
        qn, kn, vn, don = (t[a:b].float().numpy().astype(np.float64)[None] for t in (q, k, v, do))
        ro, rl = O.attention_ref(qn, kn, vn, causal=True)
        for dst, val in zip(truth, (ro,) + tuple(O.block_bwd(don, qn, kn, vn, ro, rl, None, True))):
            dst[a:b] = val[0]
    truth = [Y.extract_local_varlen(torch.from_numpy(t), cu, rank, ws, layout).float() for t in truth]
    lq, lk, lv, ldo = (Y.extract_local_varlen(t, cu, rank, ws, layout) for t in (q, k, v, do))
    for t in (lq, lk, lv):
        t.requires_grad_(True)
    cu_local = torch.tensor(cu // ws, dtype=torch.int32)
    fn = Y.zigzag_ring_flash_attn_varlen_func if impl == "zigzag" else Y.ring_flash_attn_varlen_func
    out = fn(lq, lk, lv, cu_local, int(max(lens)) // ws, causal=True, group=dist.group.WORLD)
    out.backward(ldo)
    got = [t.detach().float() for t in (out, lq.grad, lk.grad, lv.grad)]
    return all(torch.allclose(a, t, atol=tol, rtol=tol) for a, t, tol in zip(got, truth, (2e-2, 5e-2, 5e-2, 5e-2)))


@pytest.mark.parametrize("ws,impl,lens,Hq,Hkv", [(8, "zigzag", (64, 160, 32, 16), 4, 2), (8, "basic", (40, 8, 96), 2, 2),
                                                 (4, "zigzag", (8, 8, 72), 2, 1)])
def test_varlen_ring_matches_exact_attention(ws, impl, lens, Hq, Hkv):
    """Packed variable-length rings at ring degree 8 (and sequences of one 1-token chunk per rank half) against exact
    per-sequence attention, forward and backward, with GQA."""
    assert all(run_distributed(_varlen_truth_worker, ws, impl, lens, Hq, Hkv))


# ---- launches inside a ring (or a pipelined exchange) must ask for interleavable launches ---------------------
def _overlap_worker(rank, ws, ud, rd, use_async, force_groups, self_chunk="1", tails="4"):
    import yunchang_amd as Y
    import yunchang_amd.hybrid.async_attn_layer as AL0
    AL0._COMM_OVERRIDE.update(self_chunk=self_chunk, tails=tails)
    from yunchang_amd.kernels import set_block_backend
    from oracle_backend import OracleBlockBackend

    class Recording(OracleBlockBackend):
        """Mirrors HipBlockBackend's contract: `beside_transfers()` returns the backend whose launches carry
        USP_LAUNCH_INTERLEAVE; both record into one list."""

        def __init__(self, interleave=False, seen=None):
            super().__init__()
            self.interleave = interleave
            self.seen = [] if seen is None else seen
            self._beside = self if interleave else None

        def beside_transfers(self):
            if self._beside is None:
                self._beside = Recording(True, self.seen)
            return self._beside

        def fwd(self, *a, **k):
            self.seen.append(("fwd", self.interleave))
            return super().fwd(*a, **k)

        def bwd(self, *a, **k):
            self.seen.append(("bwd", self.interleave))
            return super().bwd(*a, **k)

    be = Recording()
    set_block_backend(be)
    Y.set_seq_parallel_pg(ud, rd, rank, ws)
    torch.manual_seed(0)
    q, k, v = (torch.randn(1, 64, 4, 32, dtype=torch.bfloat16).requires_grad_(True) for _ in range(3))
    if force_groups:
        import yunchang_amd.hybrid.async_attn_layer as AL
        AL._FILL_ITEMS = 1          # tiny problem: let the head-group pipeline form anyway
    layer = (Y.AsyncLongContextAttention(ring_impl_type="zigzag") if use_async
             else Y.LongContextAttention(ring_impl_type="zigzag", attn_type=Y.AttnType.HIP))
    layer(q, k, v, causal=True).sum().backward()
    return be.seen


@pytest.mark.parametrize("ud,rd,use_async,force_groups,expect,self_chunk,tails",
                         [(1, 2, False, False, True, "1", "4"),      # a ring relay is in flight
                          (2, 1, False, False, False, "0", "0"),     # one packed exchange, no self-chunk start, no tails: nothing to overlap, persistent
                          (2, 1, False, False, True, "1", "0"),      # ... with the self-chunk start (default) the group's kernels run beside ITS exchange
                          (2, 1, False, False, True, "0", "4"),      # ... with tails (default) its last pieces run beside the first pieces' exchanges
                          (2, 1, False, True, True, "1", "4"),       # head-group pipeline in LongContextAttention (default)
                          (2, 1, True, True, True, "1", "4")])       # ... and in AsyncLongContextAttention
def test_kernels_inside_a_transfer_window_are_launched_interleavable(ud, rd, use_async, force_groups, expect, self_chunk, tails):
    """Persistent launches hold every CU until they end, so a ring relay or a pipelined exchange could not
    overlap them: exactly the launches made while such transfers are in flight must come from
    `backend.beside_transfers()` (USP_LAUNCH_INTERLEAVE on the C ABI).  The backend holds no mutable state."""
    for seen in run_distributed(_overlap_worker, 2, ud, rd, use_async, force_groups, self_chunk, tails):
        assert seen and all(d == expect for _, d in seen), seen


# ---- USP_DKDV_RETURN=direct: every dK/dV block straight to its owner instead of the hop-by-hop relay ---------------
# one grid of each kind (the CPU suite runs serially in the driver): zigzag half-row blocks at ring 4, GQA beside a Ulysses
# exchange at world size 8, basic causal at ring 2 (steps that compute nothing), basic non-causal with batch 2, stripe
_RING_BWD = [f for f in MULTI if Golden(f).rd > 1 and Golden(f).bwd and any(
    tag in f for tag in ("c4_w4_u1r4", "c5_w8_u2r4_gqa_bf16", "b_w2_u1r2", "f_w4_u2r2_full_b2_gqa", "n_w4_u1r4_strip"))]


@pytest.mark.parametrize("path", _RING_BWD, ids=lambda p: p.split("/")[-1][:-4])
def test_direct_dkdv_return_is_bit_identical_to_the_relay(path, monkeypatch):
    """The owner adds the arriving blocks in step order = the order the relay adds them in: same fp32 sums, bit for
    bit, on reference grids of every kind of ring (zigzag incl. the half-row blocks, basic causal and full, stripe, batch 2,
    GQA, beside a Ulysses exchange) -- and therefore the same agreement with the reference's own run."""
    g = Golden(path)
    relay = run_distributed(_usp_worker, g.ws, path, True)
    monkeypatch.setenv("USP_DKDV_RETURN", "direct")
    direct = run_distributed(_usp_worker, g.ws, path, True)
    atol, rtol = TOL[g.dtype]["grad"]
    for r in range(g.ws):
        for key in ("out", "dq", "dk", "dv"):
            assert np.array_equal(direct[r][key], relay[r][key]), f"{g.name} {key} rank {r}"
        for key in ("dq", "dk", "dv"):
            assert_close(direct[r][key], getattr(g, key)[r], atol, rtol, f"{g.name} {key} rank {r} vs reference run")
        # the block kernels ran on the same shapes in the same order (only the transport differs)
        assert direct[r]["calls"] == relay[r]["calls"]
        assert len(direct[r]["direct"]) >= 1 and relay[r]["direct"] == []


@pytest.mark.parametrize("path", [f for f in varlen_golden_files() if "v_w4_zigzag" in f or "v_w2_basic" in f],
                         ids=lambda p: p.split("/")[-1][:-4])
def test_direct_dkdv_return_packed_rings(path, monkeypatch):
    from golden_util import VarlenGolden
    g = VarlenGolden(path)
    packed = g.Hq == g.Hkv and "oneseq" in g.name
    relay = run_distributed(_varlen_worker, g.ws, path, packed)
    monkeypatch.setenv("USP_DKDV_RETURN", "direct")
    direct = run_distributed(_varlen_worker, g.ws, path, packed)
    for r in range(g.ws):
        for key in ("out", "lse", "dq", "dk", "dv"):
            assert np.array_equal(direct[r][key], relay[r][key]), f"{g.name} {key} rank {r}"


@pytest.mark.parametrize("pieces", [2, 3])
def test_zigzag_fetch_row_ranges_at_batch_2(pieces, monkeypatch):
    """Batch 2: a wave's pieces are separate buffers (a row range of a batched buffer is not contiguous), one launch
    per source rank -- the other form of the plan -- with 2 and 3 row ranges per K/V half, against exact attention."""
    monkeypatch.setenv("USP_ZZ_PIECES", str(pieces))
    assert all(run_distributed(_async_worker, 4, 1, 4, "zigzag", 4, 2))


def _odd_views_worker(rank, ws):
    """What autograd and callers may hand the ring: an EXPANDED gradient (`out.sum().backward()`: every stride 0) and
    q/k/v views with a non-unit head-dim stride.  The device kernels take neither (unit head-dim stride, 16-byte
    multiples elsewhere -- asserted by the test backend like the C side does); the functions normalise them like
    flash-attn's maybe_contiguous does for the reference."""
    import torch.distributed as dist
    import yunchang_amd as Y
    from yunchang_amd.kernels import set_block_backend
    from oracle_backend import OracleBlockBackend
    from oracle import usp_oracle as O
    set_block_backend(OracleBlockBackend())
    Y.set_seq_parallel_pg(1, ws, rank, ws)
    torch.manual_seed(3)
    B, S, H, D = 1, 64 * ws, 2, 32
    glob = [torch.randn(B, S, H, D).to(torch.bfloat16) for _ in range(3)]
    full, _ = O.attention_ref(*(t.float().numpy().astype(np.float64) for t in glob), causal=True)
    ok = True
    for impl, fn in (("zigzag", Y.zigzag_ring_flash_attn_func), ("basic", Y.ring_flash_attn_func),
                     ("strip", Y.stripe_flash_attn_func)):
        ext = Y.EXTRACT_FUNC_DICT[impl]
        wide = []
        for t in glob:                                  # the local shard lives in every other column of a wider tensor
            loc = ext(t, rank, world_size=ws, rd=ws, ud=1)
            w = torch.zeros(loc.shape[:-1] + (2 * D,), dtype=loc.dtype)
            w[..., ::2] = loc
            wide.append(w.requires_grad_(True))
        q, k, v = (w[..., ::2] for w in wide)
        assert q.stride(-1) == 2
        out = fn(q, k, v, causal=True, group=Y.PROCESS_GROUP.RING_PG)
        out.float().sum().backward()                    # an expanded gradient reaches the ring backward
        want = ext(torch.from_numpy(full), rank, world_size=ws, rd=ws, ud=1).float()
        ok = ok and torch.allclose(out.detach().float(), want, atol=2e-2, rtol=2e-2)
        for w in wide:
            ok = ok and w.grad is not None and bool(torch.isfinite(w.grad.float()).all()) and \
                bool((w.grad[..., 1::2] == 0).all()) and bool((w.grad[..., ::2] != 0).any())
    return ok


def test_expanded_gradients_and_strided_views_are_normalised():
    assert all(run_distributed(_odd_views_worker, 2))


def test_zigzag_fetch_ungrouped_switch_gives_the_same_result(monkeypatch):
    """USP_ZZ_GROUP=0 (one launch per source rank also at batch 1: the A/B switch for the grouped launches) against the
    reference's own run of the 4-GPU grid."""
    monkeypatch.setenv("USP_ZZ_GROUP", "0")
    path = [f for f in MULTI if "c4_w4_u1r4" in f][0]
    g = Golden(path)
    res = run_distributed(_usp_worker, g.ws, path, True)
    for r in range(g.ws):
        assert_close(res[r]["out"], g.out[r], *TOL[g.dtype]["out"], f"out rank {r}")
        assert len([x for x in res[r]["calls"] if x[0] == "fwd"]) == 1 + (g.rd - 1) + (g.rd - 1 - r)


GRID = [f for f in golden_files() if Golden(f).ud > 1 and Golden(f).rd > 1 and Golden(f).layer == "hybrid"]


@pytest.mark.parametrize("mode", ["pipelined", "safe", "relay", "pipelined-pairwise", "relay-pairwise"])
@pytest.mark.parametrize("path", GRID, ids=lambda p: p.split("/")[-1][:-4])
def test_virtual_grid_runs_the_layer_for_every_rank_in_one_process(monkeypatch, path, mode):
    """tests/virtual_grid.py on host tensors (the harness the RCCL ordering test uses on the GPU): all ranks of a
    ulysses x ring grid as threads of ONE process through the layer's real autograd Function, the packed exchange
    pipelined over head groups beside the ring (two communicators' traffic interleaved) or in the safe one-exchange
    form, against the reference's goldens -- and the same sequence of collectives on every member of a group."""
    from golden_util import grad_tol
    from oracle_backend import OracleBlockBackend
    from virtual_grid import Ctx, VirtualGrid, VirtualGridPairwise, patch_dist, run_grid
    from yunchang_amd.kernels import set_block_backend
    g = Golden(path)
    # "-pairwise": no group-wide host rendezvous -- messages matched per rank pair, random host delays: the ranks drift
    pairwise = mode.endswith("-pairwise")
    mode = mode.split("-")[0]
    grid = VirtualGridPairwise(g.ud, g.rd, jitter=(7, 0.003)) if pairwise else VirtualGrid(g.ud, g.rd)
    AL = patch_dist(monkeypatch, grid)
    monkeypatch.setattr(AL, "_FILL_ITEMS", 1)
    monkeypatch.setitem(AL._COMM_OVERRIDE, "safe", mode == "safe")
    import yunchang_amd.comm.relay_exchange as RX
    monkeypatch.setitem(RX._OVERRIDE, "relay", mode == "relay")      # the pair exchange striped over the other ranks
    dtype = getattr(torch, g.dtype)
    loc = [[torch.from_numpy(np.ascontiguousarray(g.shard(x, r))).to(dtype) for x in (g.q, g.k, g.v, g.dout)]
           for r in range(g.ws)]
    prev = set_block_backend(OracleBlockBackend())
    try:
        def rank_fn(r):
            q, k, v, do = loc[r]
            upg, rpg = grid.groups_of(r)
            ctx = Ctx()
            cap = AL._MAX_GROUPS if AL.pipeline_mode(g.rd) else 1
            out = AL._AsyncUSPFunc.forward(ctx, q, k, v, None, g.causal, upg, rpg, g.impl, cap)
            grads = AL._AsyncUSPFunc.backward(ctx, do)[:3] if g.bwd else ()
            return (out,) + tuple(grads), ctx.meta[6]
        res = run_grid(grid, g.ws, rank_fn)
    finally:
        set_block_backend(prev)
    ng = {n for _, n in res}
    assert ng == ({1} if mode == "safe" else {min(AL._MAX_GROUPS, g.Hkv // g.ud)}), ng
    assert {k for k, _ in grid.calls} == ({"world", "ring"} if mode == "relay" and g.ud == 2 else {"ulysses", "ring"})
    if pairwise:          # every member of a group posted the same number of calls of each kind
        per_rank = {r: tuple(sum(1 for kk, rr in grid.calls if kk == k and rr == r) for k in ("ulysses", "world", "ring"))
                    for r in range(g.ws)}
        assert len(set(per_rank.values())) == 1, per_rank
    for r in range(g.ws):
        for t, name in zip(res[r][0], ("out", "dq", "dk", "dv")):
            tol = TOL[g.dtype]["out"] if name == "out" else grad_tol(g.dtype, g.Hq // g.Hkv if name != "dq" else 1)
            assert_close(t.float().numpy(), getattr(g, name)[r], *tol, f"{g.name} {name} rank {r}")


def _window_worker(rank, ws, ud, rd):
    """window_size through the layers at ulysses degree `ud` (ring degree 1: one block per rank after the exchange) against
    exact windowed attention on the unsharded tensors; beside a ring it must be refused, not approximated."""
    import yunchang_amd as Y
    from yunchang_amd.kernels import set_block_backend
    from oracle_backend import OracleBlockBackend
    from oracle import usp_oracle as O
    set_block_backend(OracleBlockBackend())
    Y.set_seq_parallel_pg(ud, rd, rank, ws)
    torch.manual_seed(0)
    B, S, Hq, Hkv, D, win = 2, 64 * ws, 4, 2, 32, (40, 0)
    q, k, v, do = (torch.randn(B, S, h, D).to(torch.bfloat16) for h in (Hq, Hkv, Hkv, Hq))
    ext = Y.EXTRACT_FUNC_DICT["basic"]
    lq, lk, lv, ldo = (ext(t, rank, world_size=ws, rd=rd, ud=ud).detach().clone() for t in (q, k, v, do))
    for t in (lq, lk, lv):
        t.requires_grad_(True)
    attn = Y.LongContextAttention(ring_impl_type="basic")
    if rd > 1:
        try:
            attn(lq, lk, lv, causal=True, window_size=win)
        except NotImplementedError:
            return True
        return False
    out = attn(lq, lk, lv, causal=True, window_size=win)
    out.backward(ldo)
    qn, kn, vn, don = (t.float().numpy().astype(np.float64) for t in (q, k, v, do))
    ro, rl = O.attention_ref(qn, kn, vn, causal=True, window=win)
    truth = [ext(torch.from_numpy(np.ascontiguousarray(t)), rank, world_size=ws, rd=rd, ud=ud).float()
             for t in (ro,) + tuple(O.block_bwd(don, qn, kn, vn, ro, rl, None, True, window=win))]
    got = [t.detach().float() for t in (out, lq.grad, lk.grad, lv.grad)]
    ok = all(torch.allclose(a, t, atol=tol, rtol=tol) for a, t, tol in zip(got, truth, (2e-2, 5e-2, 5e-2, 5e-2)))
    u = Y.UlyssesAttention(Y.PROCESS_GROUP.ULYSSES_PG, attn_type=Y.AttnType.HIP)
    o2 = u(lq.detach(), lk.detach(), lv.detach(), causal=True, window_size=win)
    return ok and torch.allclose(o2.float(), truth[0], atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("ws,ud,rd", [(2, 2, 1), (1, 1, 1), (2, 1, 2)])
def test_sliding_window_through_the_layers(ws, ud, rd):
    assert all(run_distributed(_window_worker, ws, ud, rd))


def _head_dim_worker(rank, ws, ud, rd, impl, D):
    """A head dim the kernels do not instantiate (96, 80, 40): the layers run it on zero-padded copies -- out and gradients
    against exact attention at the ORIGINAL head dim (softmax scale D ** -0.5)."""
    import yunchang_amd as Y
    from yunchang_amd.kernels import set_block_backend
    from oracle_backend import OracleBlockBackend
    from oracle import usp_oracle as O
    set_block_backend(OracleBlockBackend())
    Y.set_seq_parallel_pg(ud, rd, rank, ws)
    torch.manual_seed(0)
    B, S, Hq, Hkv = 1, 64 * ws, 4, 2
    q, k, v, do = (torch.randn(B, S, h, D).to(torch.bfloat16) for h in (Hq, Hkv, Hkv, Hq))
    ext = Y.EXTRACT_FUNC_DICT[impl]
    qn, kn, vn, don = (t.float().numpy().astype(np.float64) for t in (q, k, v, do))
    ro, rl = O.attention_ref(qn, kn, vn, causal=True)
    truth = [ext(torch.from_numpy(np.ascontiguousarray(t)), rank, world_size=ws, rd=rd, ud=ud).float()
             for t in (ro,) + tuple(O.block_bwd(don, qn, kn, vn, ro, rl, None, True))]
    ok = True
    for cls in (Y.LongContextAttention, Y.AsyncLongContextAttention):
        lq, lk, lv, ldo = (ext(t, rank, world_size=ws, rd=rd, ud=ud).detach().clone() for t in (q, k, v, do))
        for t in (lq, lk, lv):
            t.requires_grad_(True)
        out = cls(ring_impl_type=impl)(lq, lk, lv, causal=True)
        assert out.shape[-1] == D
        out.backward(ldo)
        got = [t.detach().float() for t in (out, lq.grad, lk.grad, lv.grad)]
        ok = ok and all(g.shape == t.shape and torch.allclose(g, t, atol=tol, rtol=tol)
                        for g, t, tol in zip(got, truth, (2e-2, 5e-2, 5e-2, 5e-2)))
    return ok


@pytest.mark.parametrize("ws,ud,rd,impl,D", [(2, 2, 1, "basic", 96), (4, 2, 2, "zigzag", 80), (1, 1, 1, "basic", 40)])
def test_head_dims_between_the_instantiated_ones_run_padded(ws, ud, rd, impl, D):
    assert all(run_distributed(_head_dim_worker, ws, ud, rd, impl, D))


def _relay_worker(rank, ws, ud, rd, low):
    """comm/relay_exchange.py: a pair's exchange striped over the other ranks (two grouped send/recv phases on the world
    group) must land the same bytes in the same places as all_to_all_single -- raw buffers with odd row counts, then the
    whole layer (forward + backward) with the relay on against the relay off, bit for bit."""
    import yunchang_amd as Y
    import yunchang_amd.comm.all_to_all as A
    import yunchang_amd.comm.relay_exchange as RX
    import yunchang_amd.hybrid.async_attn_layer as AL
    from yunchang_amd.kernels import set_block_backend
    from oracle_backend import OracleBlockBackend
    set_block_backend(OracleBlockBackend())
    Y.set_seq_parallel_pg(ud, rd, rank, ws, use_ulysses_low=low)
    upg = Y.PROCESS_GROUP.ULYSSES_PG
    ok = RX.GRID == (ud, rd, ws, low)
    for rows in (37, 64, ws + 1, 5):                       # 5 rows over 6 helpers + 2: stripes of 0 rows -> not applicable
        torch.manual_seed(rank * 100 + rows)
        send = torch.randn(2, rows, 3, 8)
        want = A._exchange(send, upg, False)
        RX._OVERRIDE["relay"] = True
        try:
            used = RX.applicable(send, upg)
            got = A._exchange(send, upg, False)
        finally:
            RX._OVERRIDE.clear()
        ok = ok and torch.equal(got, want) and used == (rows // (ws - 2 + 2) > 0)
    AL._FILL_ITEMS = 1
    torch.manual_seed(0)
    B, S, Hq, Hkv, D = 2, 32 * ws, 8, 4, 32
    q, k, v, do = (torch.randn(B, S, h, D).to(torch.bfloat16) for h in (Hq, Hkv, Hkv, Hq))
    ext = Y.EXTRACT_FUNC_DICT["zigzag"]
    res = []
    for relay in (False, True):
        RX._OVERRIDE["relay"] = relay
        lq, lk, lv, ldo = (ext(t, rank, world_size=ws, rd=rd, ud=ud).detach().clone() for t in (q, k, v, do))
        for t in (lq, lk, lv):
            t.requires_grad_(True)
        out = Y.LongContextAttention(ring_impl_type="zigzag")(lq, lk, lv, causal=True)
        out.backward(ldo)
        res.append([t.detach().clone() for t in (out, lq.grad, lk.grad, lv.grad)])
    RX._OVERRIDE.clear()
    return ok and all(torch.equal(a, b) for a, b in zip(*res))


@pytest.mark.parametrize("ws,ud,rd,low", [(4, 2, 2, True), (8, 2, 4, True), (8, 2, 4, False)])
def test_relayed_pair_exchange_is_bit_identical(ws, ud, rd, low):
    assert all(run_distributed(_relay_worker, ws, ud, rd, low))


def _relay_dp_worker(rank, ws):
    """Data-parallel x sequence-parallel (world 8 = 2 replicas of ulysses 2 x ring 2): the relayed exchange couples the
    ranks of ONE sequence-parallel block only.  The two replicas exchange buffers of DIFFERENT shapes (another batch /
    sequence length per replica is legitimate) and replica 1 makes one exchange more than replica 0: the shape agreement
    must neither raise "ranks disagree" nor wait for the other block (round 4 ran it over the world group: ADVICE.md).
    A re-initialised grid forgets the confirmed signatures."""
    import yunchang_amd as Y
    import yunchang_amd.comm.all_to_all as A
    import yunchang_amd.comm.relay_exchange as RX
    ud, rd = 2, 2
    Y.set_seq_parallel_pg(ud, rd, rank, ws)
    upg = Y.PROCESS_GROUP.ULYSSES_PG
    replica = rank // (ud * rd)
    ok = True
    RX._OVERRIDE["relay"] = True
    try:
        for n in range(2 + replica):                           # replica 1: three exchanges, replica 0: two
            rows = 16 + 8 * replica + 4 * n                    # shapes differ between the replicas and from call to call
            torch.manual_seed(rank * 10 + n)
            send = torch.randn(2, rows, 2, 8)
            RX._OVERRIDE["relay"] = False
            want = A._exchange(send, upg, False)
            RX._OVERRIDE["relay"] = True
            ok = ok and RX.applicable(send, upg) and torch.equal(A._exchange(send, upg, False), want)
        ok = ok and len(RX._AGREED) == 2 + replica
    finally:
        RX._OVERRIDE.clear()
    Y.set_seq_parallel_pg(ud, rd, rank, ws)
    return ok and len(RX._AGREED) == 0


def test_relayed_exchange_agrees_inside_its_sequence_parallel_block_only():
    assert all(run_distributed(_relay_dp_worker, 8))


def _split_steps_worker(rank, ws):
    """A ring-only grid (the 4-GPU grid: ulysses 1 x ring 4, zigzag): every backward step is issued dK/dV launch | hop posted | dQ
    launch, so the LAST hop -- which nothing else hides there -- runs beside the last step's dQ launch; bit-identical to one call
    per step (USP_BWD_SPLIT_STEPS=0)."""
    import os
    import yunchang_amd as Y
    import yunchang_amd.ring.utils as U
    from yunchang_amd.kernels import set_block_backend
    from oracle_backend import OracleBlockBackend
    be = OracleBlockBackend()
    set_block_backend(be)
    Y.set_seq_parallel_pg(1, ws, rank, ws)
    torch.manual_seed(5)
    q, k, v, do = (torch.randn(1, 64 * ws, h, 32).to(torch.bfloat16) for h in (4, 2, 2, 4))
    ext = Y.EXTRACT_FUNC_DICT["zigzag"]
    events = []
    real_commit = U.RingComm.commit
    U.RingComm.commit = lambda self: (events.append(("commit", len(self._ops))), real_commit(self))[1]
    res = []
    try:
        for split in ("1", "0"):
            os.environ["USP_BWD_SPLIT_STEPS"] = split
            lq, lk, lv, ldo = (ext(t, rank, world_size=ws, rd=ws, ud=1).detach().clone() for t in (q, k, v, do))
            for t in (lq, lk, lv):
                t.requires_grad_(True)
            out = Y.LongContextAttention(ring_impl_type="zigzag")(lq, lk, lv, causal=True)
            events.clear(); be.calls.clear()
            real_bwd = be.bwd
            be.bwd = lambda *a, **kw: (events.append(("bwd", kw.get("only"))), real_bwd(*a, **kw))[1]
            out.backward(ldo)
            be.bwd = real_bwd
            res.append([t.detach().clone() for t in (lq.grad, lk.grad, lv.grad)])
            if split == "1":
                kinds = [e for e in events if e[0] == "bwd"]
                assert [e[1] for e in kinds] == ["dkdv", "dq"] * ws, kinds
                i_last_dkdv = max(i for i, e in enumerate(events) if e == ("bwd", "dkdv"))
                i_last_dq = max(i for i, e in enumerate(events) if e == ("bwd", "dq"))
                assert any(e[0] == "commit" for e in events[i_last_dkdv:i_last_dq]), events       # the last hop is posted in between
    finally:
        U.RingComm.commit = real_commit
        os.environ.pop("USP_BWD_SPLIT_STEPS", None)
    return all(torch.equal(a, b) for a, b in zip(*res))


def test_ring_backward_steps_are_split_around_their_hop():
    assert all(run_distributed(_split_steps_worker, 4))
