"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(libusp_hip.so via yunchang_amd._C), against the CPU oracle and the reference golden fixtures.

Tolerances (stated, SURVEY.md section 8c / golden_util.TOL): bf16 out atol=rtol=2e-2, grads 5e-2;
fp16 out 4e-3, grads 1e-2; fp32 LSE 2e-3.  The reference's own bar is atol=1e-1 on the forward.
"""
import os

import numpy as np
import pytest
import torch

from golden_util import Golden, TOL, assert_close, golden_files, grad_tol, round_to
from oracle import usp_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import yunchang_amd  # noqa: F401  (fails loudly if the extension is missing)
    from yunchang_amd import _C
    _C.load()
    return torch.device("cuda:0")


def _t(x, dtype_s, dev):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(getattr(torch, dtype_s)).to(dev)


def _f(t):
    return t.detach().float().cpu().numpy()


def _rand(shape, dtype_s, seed):
    return round_to(np.random.RandomState(seed).standard_normal(shape).astype(np.float32), dtype_s)


# ------------------------------------------------------------------------------------------------
# block kernels vs oracle
# ------------------------------------------------------------------------------------------------
SHAPES = [
    # B, Sq, Sk, Hq, Hkv, D, causal, dtype
    (1, 1024, 1024, 8, 8, 64, True, "bfloat16"),      # C1
    (2, 512, 512, 4, 4, 128, True, "bfloat16"),
    (1, 384, 640, 4, 2, 128, False, "bfloat16"),      # Sq != Sk, GQA
    (1, 200, 333, 3, 1, 128, True, "bfloat16"),       # ragged, bottom-right causal
    (1, 333, 200, 2, 2, 64, True, "float16"),         # rows with no visible key
    (2, 77, 77, 2, 2, 32, True, "bfloat16"),
    (1, 1, 1, 1, 1, 128, True, "float16"),
    (1, 130, 1, 2, 1, 64, False, "bfloat16"),
]


@pytest.mark.parametrize("B,Sq,Sk,Hq,Hkv,D,causal,dt", SHAPES)
def test_block_forward_backward_vs_oracle(dev, B, Sq, Sk, Hq, Hkv, D, causal, dt):
    from yunchang_amd.kernels import hip_attn_backward, hip_attn_forward
    q, k, v, do = (_rand(s, dt, i) for i, s in enumerate(
        [(B, Sq, Hq, D), (B, Sk, Hkv, D), (B, Sk, Hkv, D), (B, Sq, Hq, D)]))
    tq, tk, tv, tdo = (_t(x, dt, dev) for x in (q, k, v, do))
    out, lse = hip_attn_forward(tq, tk, tv, 0.0, None, causal=causal)
    ro, rl = O.block_fwd(q, k, v, None, causal)
    atol, rtol = TOL[dt]["out"]
    assert_close(_f(out), ro, atol, rtol, "out")
    lse_h = _f(lse)
    fin = np.isfinite(rl)
    assert (np.isfinite(lse_h) == fin).all(), "empty rows must give lse = -inf"
    assert_close(lse_h[fin], rl[fin], 2e-3, 1e-4, "lse")
    assert (np.abs(_f(out))[~np.broadcast_to(fin.transpose(0, 2, 1)[..., None], ro.shape)] == 0).all()
    # backward with the oracle's (rounded) out / exact lse as the "global" values
    o16 = round_to(ro.astype(np.float32), dt)
    dq, dk, dv = (torch.empty_like(t) for t in (tq, tk, tv))
    hip_attn_backward(tdo, tq, tk, tv, _t(o16, dt, dev), torch.from_numpy(rl.astype(np.float32)).to(dev),
                      dq, dk, dv, 0.0, None, causal)
    rdq, rdk, rdv = O.block_bwd(do, q, k, v, o16, rl, None, causal)
    atol, rtol = TOL[dt]["grad"]
    assert_close(_f(dq), rdq, atol, rtol, "dq")
    assert_close(_f(dk), rdk, atol, rtol, "dk")
    assert_close(_f(dv), rdv, atol, rtol, "dv")


def test_strided_views_and_autograd(dev):
    """Inputs as non-contiguous (B,S,H,D) views (what the all-to-all hands over) + autograd stage."""
    from yunchang_amd.kernels import AttnType, select_flash_attn_impl
    dt = "bfloat16"
    B, S, H, D = 2, 256, 4, 64
    q, k, v, do = (_rand((B, S, H, D), dt, 10 + i) for i in range(4))
    def seq_major(x):
        return _t(x.transpose(1, 0, 2, 3), dt, dev).transpose(0, 1)      # (B,S,H,D) view of (S,B,H,D)
    tq, tk, tv = (seq_major(x).requires_grad_(True) for x in (q, k, v))
    assert not tq.is_contiguous()
    fn = select_flash_attn_impl(AttnType.HIP, "fwd-bwd")
    out = fn(tq, tk, tv, causal=True)
    out.backward(seq_major(do))
    ro, rl = O.block_fwd(q, k, v, None, True)
    rdq, rdk, rdv = O.block_bwd(do, q, k, v, ro, rl, None, True)
    assert_close(_f(out), ro, *TOL[dt]["out"], "out")
    for got, want, n in ((tq.grad, rdq, "dq"), (tk.grad, rdk, "dk"), (tv.grad, rdv, "dv")):
        assert_close(_f(got), want, *TOL[dt]["grad"], n)


def test_softmax_rescale_branch(dev):
    """Force large running-max jumps late in the KV loop (guide rule 26): one key per 64-key tile is
    aligned with the query so the max grows tile after tile."""
    from yunchang_amd.kernels import hip_attn_forward
    dt = "bfloat16"
    B, S, H, D = 1, 512, 1, 128
    rs = np.random.RandomState(3)
    q = rs.standard_normal((B, S, H, D)).astype(np.float32)
    k = rs.standard_normal((B, S, H, D)).astype(np.float32) * 0.1
    for t in range(S // 64):
        k[0, 64 * t + 5, 0] = q[0, 300, 0] * (0.5 + 0.25 * t)     # growing spikes for row 300
    v = rs.standard_normal((B, S, H, D)).astype(np.float32)
    q, k, v = (round_to(x, dt) for x in (q, k, v))
    out, lse = hip_attn_forward(_t(q, dt, dev), _t(k, dt, dev), _t(v, dt, dev), causal=False)
    ro, rl = O.block_fwd(q, k, v, None, False)
    assert_close(_f(out), ro, *TOL[dt]["out"], "out")
    assert_close(_f(lse), rl, 2e-3, 1e-4, "lse")


# ------------------------------------------------------------------------------------------------
# elementwise / copy kernels
# ------------------------------------------------------------------------------------------------
def test_merge_copy_cast_add_kernels(dev):
    from yunchang_amd import _C
    from yunchang_amd.comm import all_to_all as A
    from yunchang_amd.ring.utils import update_out_and_lse
    dt = "bfloat16"
    B, S, H, D = 2, 96, 4, 64
    rs = np.random.RandomState(0)
    # update_out_and_lse (ring/utils.py:10-51) incl. the slice form
    bo1, bo2 = (_rand((B, S, H, D), dt, 20 + i) for i in range(2))
    bl1, bl2 = (rs.standard_normal((B, H, S)).astype(np.float32) for _ in range(2))
    out, lse = update_out_and_lse(None, None, _t(bo1, dt, dev), torch.from_numpy(bl1).to(dev))
    out, lse = update_out_and_lse(out, lse, _t(bo2, dt, dev), torch.from_numpy(bl2).to(dev))
    ro, rlse = O.update_out_and_lse(None, None, bo1, bl1)
    ro, rlse = O.update_out_and_lse(ro, rlse, bo2, bl2)
    assert_close(_f(out), ro, 1e-5, 1e-5, "merged out")
    assert_close(_f(lse), rlse, 1e-5, 1e-5, "merged lse")
    half = S // 2
    bo3 = _rand((B, half, H, D), dt, 30)
    bl3 = rs.standard_normal((B, H, half)).astype(np.float32)
    out, lse = update_out_and_lse(out, lse, _t(bo3, dt, dev), torch.from_numpy(bl3).to(dev),
                                  slice_=(slice(None), slice(half, None)))
    ro, rlse = O.update_out_and_lse(ro, rlse, bo3, bl3, row_slice=slice(half, None))
    assert_close(_f(out), ro, 1e-5, 1e-5, "slice-merged out")
    assert_close(_f(lse), rlse, 1e-5, 1e-5, "slice-merged lse")
    # pack / unpack of the Ulysses exchange (bit exact)
    P = 2
    x = torch.randn(B, S, H, D, device=dev).to(torch.bfloat16)
    send = A.pack_heads(x, P)
    hp = H // P
    want = x.reshape(B, S, P, hp, D).permute(2, 1, 0, 3, 4).contiguous()
    assert torch.equal(send, want)
    y = A.view_seq(send)                                   # treat as if received
    assert y.shape == (B, P * S, hp, D)
    back = A.unpack_heads(A.pack_seq(x[:, :, :hp].contiguous(), P))
    want2 = x[:, :, :hp].reshape(B, P, S // P, hp, D).permute(0, 2, 1, 3, 4).reshape(B, S // P, H, D)
    assert torch.equal(back, want2)
    # packed-qkv (5-D) exchange buffers, B > 1 (bit exact)
    x5 = torch.randn(B, S, 3, H, D, device=dev).to(torch.bfloat16)
    send5 = A.pack_heads_5d(x5, P)
    want5 = x5.reshape(B, S, 3, P, hp, D).permute(3, 1, 0, 2, 4, 5).contiguous()
    assert torch.equal(send5, want5)
    seq5 = A.view_seq_5d(send5)
    assert seq5.shape == (B, P * S, 3, hp, D)
    y5 = x5[:, :, :, :hp].contiguous()                       # (B, S, 3, hp, D) as if produced locally
    back5 = A.unpack_heads_5d(A.pack_seq_5d(y5, P))
    want5b = y5.reshape(B, P, S // P, 3, hp, D).permute(0, 2, 3, 1, 4, 5).reshape(B, S // P, 3, H, D)
    assert torch.equal(back5, want5b)
    # cast + add (bit exact vs torch)
    a = torch.randn(B, S, H, D, device=dev)
    b = torch.randn(B, S, H, D, device=dev)
    d16 = torch.empty(B, S, H, D, device=dev, dtype=torch.bfloat16)
    _C.cast_from_f32(d16, a)
    assert torch.equal(d16, a.to(torch.bfloat16))
    d16h = torch.empty(B, S, H, D, device=dev, dtype=torch.float16)
    _C.cast_from_f32(d16h, a)
    assert torch.equal(d16h, a.to(torch.float16))
    c = torch.empty_like(a)
    _C.add_f32(c, a, b)
    assert torch.equal(c, a + b)
    _C.add_f32(a[:, :half], a[:, :half], b[:, :half])       # in place on batch slices
    assert torch.equal(a[:, :half], (c - b + b)[:, :half]) or torch.allclose(a[:, :half], c[:, :half])


def test_unsupported_arguments_raise(dev):
    from yunchang_amd import _C
    from yunchang_amd.kernels import hip_attn_forward
    q = torch.randn(1, 64, 2, 96, device=dev, dtype=torch.bfloat16)
    lse = torch.empty((1, 2, 64), dtype=torch.float32, device=dev)
    with pytest.raises(RuntimeError, match="unsupported"):      # the C ABI instantiates head dims 32 / 64 / 128 ...
        _C.flash_fwd(q, q, q, 0.1, False, lse, torch.empty_like(q))
    assert hip_attn_forward(q, q, q)[0].shape == q.shape        # ... the entry points serve the ones below 128 padded
    big = torch.randn(1, 64, 2, 256, device=dev, dtype=torch.bfloat16)
    with pytest.raises(NotImplementedError):
        hip_attn_forward(big, big, big)
    q32 = torch.randn(1, 64, 2, 64, device=dev)
    with pytest.raises(TypeError):
        hip_attn_forward(q32, q32, q32)


# ------------------------------------------------------------------------------------------------
# golden fixtures from the reference: single-rank through the real entry point
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def single_rank_pg(dev):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29731")
    if not dist.is_initialized():
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)
    import yunchang_amd as Y
    Y.set_seq_parallel_pg(1, 1, 0, 1)
    yield
    dist.destroy_process_group()


W1 = [f for f in golden_files() if "_w1_" in f and "fp32" not in f]


@pytest.mark.parametrize("path", W1, ids=lambda p: p.split("/")[-1][:-4])
def test_single_rank_golden_through_long_context_attention(dev, single_rank_pg, path):
    import yunchang_amd as Y
    g = Golden(path)
    tq, tk, tv, tdo = (_t(x, g.dtype, dev) for x in (g.q, g.k, g.v, g.dout))
    if g.bwd:
        for t in (tq, tk, tv):
            t.requires_grad_(True)
    attn = Y.LongContextAttention(ring_impl_type=g.impl, attn_type=Y.AttnType.TORCH_EFFICIENT)
    out = attn(tq, tk, tv, dropout_p=0, causal=True, window_size=(-1, -1), softcap=0.0,
               alibi_slopes=None, deterministic=False, return_attn_probs=True)
    assert out.shape == tq.shape and out.dtype == tq.dtype
    assert_close(_f(out), g.out[0], *TOL[g.dtype]["out"], f"{g.name} out vs reference run")
    if g.bwd:
        out.backward(tdo)
        for key, t in (("dq", tq), ("dk", tk), ("dv", tv)):
            assert_close(_f(t.grad), getattr(g, key)[0], *TOL[g.dtype]["grad"], f"{g.name} {key}")


def test_async_layer_on_gpu_streams(dev, single_rank_pg):
    """AsyncLongContextAttention on the real side-stream lane (1-rank RCCL group: exchanges are
    self-exchanges, 4 head groups): same result as LongContextAttention, forward and backward."""
    import yunchang_amd as Y
    import yunchang_amd.hybrid.async_attn_layer as AL
    B, S, Hq, Hkv, D = 2, 1024, 8, 4, 128
    groups = AL._groups
    AL._groups = lambda hq, hkv, P, B=None, S=None, max_groups=None, link_bound=False, k_split=False: (4, hkv // P // 4, hq // hkv)   # 4 groups at P = 1
    request_restore = groups
    gen = torch.Generator(device="cpu").manual_seed(5)
    q, k, v, do = (torch.randn(B, S, h, D, generator=gen).to(torch.bfloat16).to(dev) for h in (Hq, Hkv, Hkv, Hq))
    res = []
    for cls in (Y.LongContextAttention, Y.AsyncLongContextAttention):
        tq, tk, tv = (t.clone().requires_grad_(True) for t in (q, k, v))
        for _ in range(3):                      # repeat: exposes stream / allocator races
            tq.grad = tk.grad = tv.grad = None
            out = cls(ring_impl_type="zigzag")(tq, tk, tv, causal=True)
            out.backward(do)
        torch.cuda.synchronize()
        res.append([out.detach(), tq.grad, tk.grad, tv.grad])
    AL._groups = request_restore
    for a, b, n in zip(res[0], res[1], ("out", "dq", "dk", "dv")):
        assert torch.allclose(a.float(), b.float(), atol=4e-3, rtol=4e-3), n
    # ... and against exact attention (fp64 oracle), not only against the sibling layer
    qn, kn, vn, don = (_f(t).astype(np.float64) for t in (q, k, v, do))
    ro, rl = O.attention_ref(qn, kn, vn, causal=True)
    truth = (ro,) + tuple(O.block_bwd(don, qn, kn, vn, ro, rl, None, True))
    for layer_res in res:
        for a, t, n in zip(layer_res, truth, ("out", "dq", "dk", "dv")):
            assert_close(_f(a), t, *TOL["bfloat16"]["out" if n == "out" else "grad"], f"async lane {n} vs oracle")


def test_c1_fp32_fixture_is_matched_by_bf16_kernel(dev, single_rank_pg):
    """BASELINE configs[0] (the reference's CPU-runnable case, fp32): our 16-bit kernel on the
    bf16-rounded inputs must sit within the bf16 envelope of the reference's fp32 result."""
    import yunchang_amd as Y
    g = Golden([f for f in golden_files() if "c1_w1_fp32" in f][0])
    tq, tk, tv = (_t(x, "bfloat16", dev) for x in (g.q, g.k, g.v))
    out = Y.LongContextAttention(ring_impl_type="basic")(tq, tk, tv, causal=True)
    assert_close(_f(out), g.out[0], *TOL["bfloat16"]["out"], "C1 out vs reference fp32 run")


# ------------------------------------------------------------------------------------------------
# golden fixtures from the reference: multi-rank grids emulated with VIRTUAL RANKS on one GPU
# (the package's real pack/unpack kernels and ring step functions; only the wire is emulated)
# ------------------------------------------------------------------------------------------------
MULTI = [f for f in golden_files() if "_w1_" not in f]


def _virtual_usp(g, dev, with_bwd):
    from yunchang_amd.comm import all_to_all as A
    from yunchang_amd.kernels import get_block_backend
    from yunchang_amd.ring import ring_flash_attn as RB
    from yunchang_amd.ring import stripe_flash_attn as RS
    from yunchang_amd.ring import zigzag_ring_flash_attn as RZ
    be = get_block_backend()
    assert be.name == "hip"
    ws, ud, rd, dt = g.ws, g.ud, g.rd, g.dtype
    tdt = getattr(torch, dt)
    ulysses, ring = O.seq_parallel_groups(ud, rd, ws)
    scale = g.D ** -0.5
    loc = {n: [_t(g.shard(getattr(g, n), r), dt, dev) for r in range(ws)] for n in ("q", "k", "v", "dout")}

    def exchange_heads(xs):           # xs: per-rank (B,Sl,H,D) -> per-rank (B,S,H/P,D)
        res = [None] * ws
        for grp in ulysses:
            P = len(grp)
            if P == 1:
                res[grp[0]] = xs[grp[0]]
                continue
            sends = [A.pack_heads(xs[r], P) for r in grp]
            for i, r in enumerate(grp):
                recv = torch.stack([sends[j][i] for j in range(P)])
                res[r] = A.view_seq(recv)
        return res

    def exchange_seq(xs):             # per-rank (B,S,H/P,D) -> per-rank (B,Sl,H,D)
        res = [None] * ws
        for grp in ulysses:
            P = len(grp)
            if P == 1:
                res[grp[0]] = xs[grp[0]]
                continue
            sends = [A.pack_seq(xs[r], P) for r in grp]
            for i, r in enumerate(grp):
                recv = torch.stack([sends[j][i] for j in range(P)])
                res[r] = A.unpack_heads(recv)
        return res

    def exchange_5d(xs, pack, unpack):   # packed qkv (B,*,3,H,D): SeqAllToAll5D's kernels
        res = [None] * ws
        for grp in ulysses:
            P = len(grp)
            if P == 1:
                res[grp[0]] = xs[grp[0]]
                continue
            sends = [pack(xs[r], P) for r in grp]
            for i, r in enumerate(grp):
                res[r] = unpack(torch.stack([sends[j][i] for j in range(P)]))
        return res

    if g.layer == "qkvpacked":           # ONE exchange of the stacked tensor; q, k, v are views of its result
        qkv = [torch.stack([loc["q"][r], loc["k"][r], loc["v"][r]], dim=2) for r in range(ws)]
        h5 = exchange_5d(qkv, A.pack_heads_5d, A.view_seq_5d)
        hq, hk, hv = ([t[:, :, i] for t in h5] for i in range(3))
    else:
        hq, hk, hv = exchange_heads(loc["q"]), exchange_heads(loc["k"]), exchange_heads(loc["v"])
    outs, lses = [None] * ws, [None] * ws
    for grp in ring:
        P = len(grp)
        for r, rank in enumerate(grp):
            q = hq[rank]
            B, S, H, D = q.shape
            out = torch.empty((B, S, H, D), dtype=tdt, device=dev)
            lse = torch.empty((B, H, S), dtype=torch.float32, device=dev)
            acc = torch.empty((B, S, H, D), dtype=torch.float32, device=dev) if P > 1 else None
            for step in range(P):
                src = grp[(r - step) % P]
                if g.impl == "zigzag":
                    RZ.zigzag_fwd_step(be, r, P, step, q, hk[src], hv[src], scale, lse, out, acc)
                elif g.impl == "strip":
                    RS.stripe_fwd_step(be, r, P, step, q, hk[src].contiguous(), hv[src].contiguous(), scale,
                                       lse, out, acc)
                else:
                    RB.basic_fwd_step(be, r, P, step, g.causal, q, hk[src], hv[src], scale, lse, out, acc)
            outs[rank], lses[rank] = out, lse
    final = exchange_seq(outs)
    if not with_bwd:
        return final, None
    hdo = exchange_heads(loc["dout"])
    f32 = torch.float32
    hdq, hdk, hdv = [None] * ws, [None] * ws, [None] * ws
    for grp in ring:
        P = len(grp)
        st = []
        for r, rank in enumerate(grp):
            q = hq[rank]
            B, S, H, D = q.shape
            delta = torch.empty((B, H, S), dtype=f32, device=dev)
            be.delta(hdo[rank], outs[rank], delta)
            st.append(dict(delta=delta, dq=torch.empty((B, S, H, D), dtype=f32, device=dev),
                           dk=torch.empty(hk[rank].shape, dtype=f32, device=dev),
                           dv=torch.empty(hv[rank].shape, dtype=f32, device=dev),
                           bk=torch.empty(hk[rank].shape, dtype=f32, device=dev),
                           bv=torch.empty(hv[rank].shape, dtype=f32, device=dev)))
        c = hq[grp[0]].shape[1] // 2
        for step in range(P):
            if step > 0:                                       # the wire: accumulators move to rank+1
                dks, dvs = [s["dk"] for s in st], [s["dv"] for s in st]
                for r in range(P):
                    st[r]["dk"], st[r]["dv"] = dks[(r - 1) % P], dvs[(r - 1) % P]
            for r, rank in enumerate(grp):
                s = st[r]
                src = grp[(r - step) % P]
                dst_k, dst_v = (s["dk"], s["dv"]) if step == 0 else (s["bk"], s["bv"])
                if g.impl == "zigzag":
                    RZ.zigzag_bwd_block(be, r, P, step, hdo[rank], hq[rank], hk[src], hv[src], lses[rank],
                                        s["delta"], scale, s["dq"], dst_k, dst_v)
                    if step > 0:
                        RZ.zigzag_bwd_fold(be, r, step, c, s["dk"], s["dv"], s["bk"], s["bv"])
                elif g.impl == "strip":
                    RS.stripe_bwd_block(be, r, P, step, hdo[rank], hq[rank], hk[src].contiguous(),
                                        hv[src].contiguous(), lses[rank], s["delta"], scale, s["dq"], dst_k, dst_v)
                    if step > 0:
                        RS.stripe_bwd_fold(be, r, step, s["dk"], s["dv"], s["bk"], s["bv"])
                else:
                    did = RB.basic_bwd_block(be, r, P, step, g.causal, hdo[rank], hq[rank], hk[src], hv[src],
                                             lses[rank], s["delta"], scale, s["dq"], dst_k, dst_v)
                    if step > 0 and did:
                        be.add(s["dk"], s["dk"], s["bk"])
                        be.add(s["dv"], s["dv"], s["bv"])
        dks, dvs = [s["dk"] for s in st], [s["dv"] for s in st]
        for r, rank in enumerate(grp):                         # final hop lands on the owner
            hdq[rank] = st[r]["dq"].to(tdt)
            hdk[rank] = dks[(r - 1) % P].to(tdt) if P > 1 else dks[r].to(tdt)
            hdv[rank] = dvs[(r - 1) % P].to(tdt) if P > 1 else dvs[r].to(tdt)
    if g.layer == "qkvpacked":           # the gradient of the packed exchange is the inverse packed exchange
        g5 = exchange_5d([torch.stack([hdq[r], hdk[r], hdv[r]], dim=2) for r in range(ws)],
                         A.pack_seq_5d, A.unpack_heads_5d)
        return final, tuple([t[:, :, i] for t in g5] for i in range(3))
    return final, (exchange_seq(hdq), exchange_seq(hdk), exchange_seq(hdv))


@pytest.mark.parametrize("path", MULTI, ids=lambda p: p.split("/")[-1][:-4])
def test_multi_rank_golden_with_virtual_ranks(dev, path):
    g = Golden(path)
    outs, grads = _virtual_usp(g, dev, g.bwd)
    for r in range(g.ws):
        assert_close(_f(outs[r]), g.out[r], *TOL[g.dtype]["out"], f"{g.name} out rank {r}")
    if g.bwd:
        for name, per_rank in zip(("dq", "dk", "dv"), grads):
            for r in range(g.ws):
                assert_close(_f(per_rank[r]), getattr(g, name)[r], *TOL[g.dtype]["grad"],
                             f"{g.name} {name} rank {r}")


# ------------------------------------------------------------------------------------------------
# BASELINE full sizes: size-independent properties + sampled oracle rows
# ------------------------------------------------------------------------------------------------
def _sampled_rows_oracle(q, k, v, b, h, rows, causal, g):
    """Exact attention for a few query rows of one head (numpy fp64)."""
    qs = q[b, rows, h].astype(np.float64)                       # (R,D)
    kk = k[b, :, h // g].astype(np.float64)
    vv = v[b, :, h // g].astype(np.float64)
    s = qs @ kk.T * (q.shape[-1] ** -0.5)
    if causal:
        off = k.shape[1] - q.shape[1]
        s = np.where(np.arange(k.shape[1])[None, :] > np.asarray(rows)[:, None] + off, -np.inf, s)
    m = s.max(-1, keepdims=True)
    p = np.exp(s - m)
    l = p.sum(-1, keepdims=True)
    return (p / l) @ vv, (m + np.log(l))[:, 0]


def test_c2_full_size_properties(dev):
    """BASELINE configs[1]: B=2 S=8192 H=16 D=128 bf16 causal, on the real kernel."""
    from yunchang_amd import _C
    from yunchang_amd.kernels import hip_attn_forward
    B, S, H, D = 2, 8192, 16, 128
    gen = torch.Generator(device="cpu").manual_seed(0)
    q, k, v = (torch.randn(B, S, H, D, generator=gen).to(torch.bfloat16) for _ in range(3))
    tq, tk, tv = q.to(dev), k.to(dev), v.to(dev)
    out, lse = hip_attn_forward(tq, tk, tv, causal=True)
    # (1) sampled rows of two heads against exact attention
    qn, kn, vn = (t.float().numpy() for t in (q, k, v))
    rows = [0, 1, 63, 64, 255, 256, 257, 4095, 4096, 8000, 8191]
    for (b, h) in ((0, 0), (1, 13)):
        ro, rl = _sampled_rows_oracle(qn, kn, vn, b, h, rows, True, 1)
        assert_close(_f(out[b, rows, h]), ro, *TOL["bfloat16"]["out"], f"sampled out b{b} h{h}")
        assert_close(_f(lse[b, h, rows]), rl, 2e-3, 1e-4, f"sampled lse b{b} h{h}")
    # (2) linearity in V: attention(q, k, 2v) == 2 attention(q, k, v) exactly (power-of-two scale)
    out2, lse2 = hip_attn_forward(tq, tk, tv * 2, causal=True)
    assert torch.equal(out2.float(), out.float() * 2) and torch.equal(lse2, lse)
    # (3) causal prefix property: rows [0, S/2) do not depend on later keys
    outp, lsep = hip_attn_forward(tq[:, : S // 2], tk[:, : S // 2], tv[:, : S // 2], causal=True)
    assert torch.allclose(outp.float(), out[:, : S // 2].float(), atol=1e-2, rtol=1e-2)
    assert torch.allclose(lsep, lse[:, :, : S // 2], atol=1e-4, rtol=1e-5)
    # (4) KV-split + fused merge == one pass (non-causal), i.e. what a ring of 2 computes
    full, lse_full = hip_attn_forward(tq, tk, tv, causal=False)
    acc = torch.empty(B, S, H, D, device=dev, dtype=torch.float32)
    lse_m = torch.empty(B, H, S, device=dev, dtype=torch.float32)
    outm = torch.empty_like(full)
    hs = S // 2
    _C.flash_fwd(tq, tk[:, :hs], tv[:, :hs], D ** -0.5, False, lse_m, outm, acc, False, 0, 0)
    _C.flash_fwd(tq, tk[:, hs:], tv[:, hs:], D ** -0.5, False, lse_m, outm, acc, True, 0, S)
    assert torch.allclose(outm.float(), full.float(), atol=8e-3, rtol=8e-3)
    assert torch.allclose(lse_m, lse_full, atol=1e-4, rtol=1e-5)
    # (5) softmax rows are convex combinations: |out| <= max |v| per head
    vmax = tv.float().abs().amax(dim=1, keepdim=True)
    assert (out.float().abs() <= vmax + 1e-2).all()


def test_c5_rank_block_gqa_backward_sampled(dev):
    """The per-rank block of BASELINE configs[4] after the Ulysses exchange (Hq=16, Hkv=2, D=128),
    at a reduced sequence (S=2048): GQA backward against the oracle on one kv head."""
    from yunchang_amd.kernels import hip_attn_backward, hip_attn_forward
    dt = "bfloat16"
    B, S, Hq, Hkv, D = 1, 2048, 16, 2, 128
    q, k, v, do = (_rand(s, dt, 40 + i) for i, s in enumerate(
        [(B, S, Hq, D), (B, S, Hkv, D), (B, S, Hkv, D), (B, S, Hq, D)]))
    tq, tk, tv, tdo = (_t(x, dt, dev) for x in (q, k, v, do))
    out, lse = hip_attn_forward(tq, tk, tv, causal=True)
    dq, dk, dv = (torch.empty_like(t) for t in (tq, tk, tv))
    hip_attn_backward(tdo, tq, tk, tv, out, lse, dq, dk, dv, 0.0, None, True)
    g = Hq // Hkv
    sl = slice(g, 2 * g)                                     # query heads of kv head 1
    ro, rl = O.block_fwd(q[:, :, sl], k[:, :, 1:2], v[:, :, 1:2], None, True)
    rdq, rdk, rdv = O.block_bwd(do[:, :, sl], q[:, :, sl], k[:, :, 1:2], v[:, :, 1:2],
                                _f(out)[:, :, sl], _f(lse)[:, sl], None, True)
    assert_close(_f(out)[:, :, sl], ro, *TOL[dt]["out"], "out")
    assert_close(_f(dq)[:, :, sl], rdq, *TOL[dt]["grad"], "dq")
    # dK / dV sum 8 query heads: absolute tolerance scaled by sqrt(8) (golden_util.grad_tol has the derivation) ...
    assert_close(_f(dk)[:, :, 1:2], rdk, *grad_tol(dt, g), "dk (sum over 8 query heads)")
    assert_close(_f(dv)[:, :, 1:2], rdv, *grad_tol(dt, g), "dv (sum over 8 query heads)")
    # ... and the scale-free criterion of SURVEY 8(c): our error against the fp64 truth is at most twice the error
    # of the 16-bit third-party path the reference would run here (torch SDPA autograd, K/V heads expanded)
    import torch.nn.functional as F
    ref = [tq[:, :, sl].transpose(1, 2).clone().requires_grad_(True),
           tk[:, :, 1:2].expand(B, S, g, D).transpose(1, 2).clone().requires_grad_(True),
           tv[:, :, 1:2].expand(B, S, g, D).transpose(1, 2).clone().requires_grad_(True)]
    try:
        F.scaled_dot_product_attention(*ref, is_causal=True).backward(tdo[:, :, sl].transpose(1, 2))
    except Exception as e:                     # pragma: no cover - depends on the torch build
        pytest.skip(f"torch SDPA backward does not run here: {e!r}")
    for name, ours, theirs, truth in (("dk", _f(dk)[:, :, 1], _f(ref[1].grad.float().sum(1)), rdk[:, :, 0]),
                                      ("dv", _f(dv)[:, :, 1], _f(ref[2].grad.float().sum(1)), rdv[:, :, 0])):
        e_ours, e_ref = np.abs(ours - truth).max(), np.abs(theirs - truth).max()
        assert e_ours <= 2 * e_ref + 1e-3, f"{name}: our max err {e_ours:.3e} vs reference path {e_ref:.3e}"


# ------------------------------------------------------------------------------------------------
# packed variable-length mode (include/usp_hip.h seq_q / seq_k): kernels vs oracle, then the varlen ring
# schedules vs the reference's own runs with virtual ranks
# ------------------------------------------------------------------------------------------------
PACKED = [
    # lens_q, lens_k (None = same), Hq, Hkv, D, causal, dtype
    ((64, 192, 320, 8), None, 4, 2, 128, True, "bfloat16"),      # GQA -> head-split workspace path
    ((130, 62, 257, 1), None, 2, 2, 64, True, "bfloat16"),       # ragged, single-row sequence
    ((100, 300), (260, 40), 2, 1, 128, False, "float16"),        # different q / k lengths, full
    ((96, 33), (200, 77), 3, 3, 32, True, "bfloat16"),           # bottom-right causal per sequence
]


def _tables(lens, dev):
    first = np.concatenate([[0], np.cumsum(lens)[:-1]])
    return torch.tensor(np.stack([first, lens], 1), dtype=torch.int32, device=dev)


@pytest.mark.parametrize("lq,lk,Hq,Hkv,D,causal,dt", PACKED)
def test_packed_kernels_vs_oracle(dev, lq, lk, Hq, Hkv, D, causal, dt):
    from yunchang_amd import _C
    lk = lq if lk is None else lk
    Tq, Tk = sum(lq), sum(lk)
    q, k, v, do = (_rand(s, dt, 10 + i) for i, s in enumerate(
        [(Tq, Hq, D), (Tk, Hkv, D), (Tk, Hkv, D), (Tq, Hq, D)]))
    tq, tk, tv, tdo = (_t(x, dt, dev) for x in (q, k, v, do))
    sq, sk = _tables(lq, dev), _tables(lk, dev)
    scale = D ** -0.5
    out = torch.full((Tq, Hq, D), float("nan"), dtype=tq.dtype, device=dev)
    lse = torch.full((Hq, Tq), float("nan"), dtype=torch.float32, device=dev)
    _C.flash_fwd_packed(tq, tk, tv, sq, sk, max(lq), max(lk), scale, causal, lse, out=out)
    cq, ck = np.concatenate([[0], np.cumsum(lq)]), np.concatenate([[0], np.cumsum(lk)])
    ro = np.zeros((Tq, Hq, D)); rl = np.zeros((Hq, Tq))
    rdq = np.zeros((Tq, Hq, D)); rdk = np.zeros((Tk, Hkv, D)); rdv = np.zeros((Tk, Hkv, D))
    for i in range(len(lq)):
        a, b, c, d = cq[i], cq[i + 1], ck[i], ck[i + 1]
        o_i, l_i = O.block_fwd(q[None, a:b], k[None, c:d], v[None, c:d], scale, causal)
        ro[a:b], rl[:, a:b] = o_i[0], l_i[0]
    atol, rtol = TOL[dt]["out"]
    assert_close(_f(out), ro, atol, rtol, "packed out")
    fin = np.isfinite(rl)
    lse_h = _f(lse)
    assert (np.isfinite(lse_h) == fin).all()
    assert_close(lse_h[fin], rl[fin], 2e-3, 1e-4, "packed lse")
    # backward, fed with the oracle's rounded out / exact lse
    o16 = round_to(ro.astype(np.float32), dt)
    for i in range(len(lq)):
        a, b, c, d = cq[i], cq[i + 1], ck[i], ck[i + 1]
        g = O.block_bwd(do[None, a:b], q[None, a:b], k[None, c:d], v[None, c:d], o16[None, a:b],
                        rl[None, :, a:b], scale, causal)
        rdq[a:b], rdk[c:d], rdv[c:d] = g[0][0], g[1][0], g[2][0]
    delta = torch.empty((Hq, Tq), dtype=torch.float32, device=dev)
    _C.bwd_delta(tdo[None], _t(o16, dt, dev)[None], delta[None])
    lse_t = torch.from_numpy(rl.astype(np.float32)).to(dev)
    dq, dk, dv = (torch.empty_like(t) for t in (tq, tk, tv))
    _C.flash_bwd_packed(tdo, tq, tk, tv, lse_t, delta, sq, sk, max(lq), max(lk), None, None, None, scale,
                        causal, dq16=dq, dk16=dk, dv16=dv)
    atol, rtol = TOL[dt]["grad"]
    assert_close(_f(dq), rdq, atol, rtol, "packed dq")
    assert_close(_f(dk), rdk, atol, rtol, "packed dk")
    assert_close(_f(dv), rdv, atol, rtol, "packed dv")


def test_packed_half_tables_merge_and_accumulate(dev):
    """Sub-range tables (the zigzag s<=r / s>r steps): rows outside the tables must not be touched, the
    fused merge must equal the oracle merge, fp32 gradients must accumulate in place."""
    from yunchang_amd import _C
    from yunchang_amd.ring.varlen_utils import SeqTables
    dt, H, D = "bfloat16", 2, 64
    lens = (64, 200, 36)
    T = sum(lens)
    q, k, v, do = (_rand((T, H, D), dt, 20 + i) for i in range(4))
    tq, tk, tv, tdo = (_t(x, dt, dev) for x in (q, k, v, do))
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=dev)
    tb = SeqTables(cu, max(lens), dev)
    scale = D ** -0.5
    # step A: q x front halves (full), adopt; step B: back halves of q x everything, merged in place
    acc = torch.full((T, H, D), float("nan"), dtype=torch.float32, device=dev)
    out = torch.full((T, H, D), float("nan"), dtype=tq.dtype, device=dev)
    lse = torch.full((H, T), float("nan"), dtype=torch.float32, device=dev)
    _C.flash_fwd_packed(tq, tk, tv, tb.full, tb.front, tb.max_full, tb.max_half, scale, False, lse, out, acc,
                        False, 0, 1)                        # front half rows final, back half rows -> acc
    _C.flash_fwd_packed(tq, tk, tv, tb.back, tb.full, tb.max_half, tb.max_full, scale, False, lse, out, acc,
                        True, 0, 2)
    ro = np.zeros((T, H, D)); rl = np.zeros((H, T))
    cs = np.concatenate([[0], np.cumsum(lens)])
    for i, n in enumerate(lens):
        a, h = cs[i], n // 2
        o1, l1 = O.block_fwd(q[None, a:a + n], k[None, a:a + h], v[None, a:a + h], scale, False)
        o2, l2 = O.block_fwd(q[None, a + h:a + n], k[None, a:a + n], v[None, a:a + n], scale, False)
        o, l_ = O.update_out_and_lse(None, None, o1, l1)
        o, l_ = O.update_out_and_lse(o, l_, o2, l2, row_slice=slice(h, None))
        ro[a:a + n], rl[:, a:a + n] = o[0], l_[0, :, :, 0].T
    assert_close(_f(out), ro, *TOL[dt]["out"], "merged out")
    assert_close(_f(lse), rl, 2e-3, 1e-4, "merged lse")
    # backward into front-half key rows only, accumulating on top of known values
    delta = torch.zeros((H, T), dtype=torch.float32, device=dev)
    lse_t = torch.from_numpy(rl.astype(np.float32)).to(dev)
    base = 3.0
    dq = torch.full((T, H, D), base, dtype=torch.float32, device=dev)
    dk = torch.full((T, H, D), base, dtype=torch.float32, device=dev)
    dv = torch.full((T, H, D), base, dtype=torch.float32, device=dev)
    _C.flash_bwd_packed(tdo, tq, tk, tv, lse_t, delta, tb.full, tb.front, tb.max_full, tb.max_half, dq, dk, dv,
                        scale, False, accum_dq=True, accum_dk=True, accum_dv=True)
    rdq = np.full((T, H, D), base); rdk = np.full((T, H, D), base); rdv = np.full((T, H, D), base)
    for i, n in enumerate(lens):
        a, h = cs[i], n // 2
        zero_out = np.zeros((1, n, H, D))                      # delta = 0  <=>  out = 0
        g = O.block_bwd(do[None, a:a + n], q[None, a:a + n], k[None, a:a + h], v[None, a:a + h], zero_out,
                        rl[None, :, a:a + n], scale, False)
        rdq[a:a + n] += g[0][0]; rdk[a:a + h] += g[1][0]; rdv[a:a + h] += g[2][0]
    atol, rtol = TOL[dt]["grad"]
    assert_close(_f(dq), rdq, atol, rtol, "dq")
    assert_close(_f(dk), rdk, atol, rtol, "dk (back-half rows must stay untouched)")
    assert_close(_f(dv), rdv, atol, rtol, "dv (back-half rows must stay untouched)")


from golden_util import VarlenGolden, varlen_golden_files  # noqa: E402


@pytest.mark.parametrize("path", varlen_golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_varlen_ring_golden_with_virtual_ranks(dev, path):
    """The package's packed ring step functions on the HIP kernels, the ring emulated with virtual ranks on
    one GPU, against the reference's (zigzag_)ring_flash_attn_varlen_func run (tests/golden/v_*.npz)."""
    from yunchang_amd.kernels import get_block_backend
    from yunchang_amd.ring import ring_flash_attn_varlen as RB
    from yunchang_amd.ring import zigzag_ring_flash_attn_varlen as RZ
    from yunchang_amd.ring.varlen_utils import SeqTables
    g = VarlenGolden(path)
    be = get_block_backend()
    assert be.name == "hip"
    P, dt = g.ws, g.dtype
    tdt, f32 = getattr(torch, dt), torch.float32
    scale = g.D ** -0.5
    loc = {n: [_t(g.shard(getattr(g, n), r), dt, dev) for r in range(P)] for n in ("q", "k", "v", "dout")}
    tb = SeqTables(torch.tensor(g.cu_local, dtype=torch.int32), g.max_local, dev)
    T = loc["q"][0].shape[0]
    outs, lses = [], []
    for r in range(P):
        out = torch.empty((T, g.Hq, g.D), dtype=tdt, device=dev)
        lse = torch.empty((g.Hq, T), dtype=f32, device=dev)
        acc = torch.empty((T, g.Hq, g.D), dtype=f32, device=dev)
        for step in range(P):
            src = (r - step) % P
            if g.impl == "zigzag":
                RZ.zigzag_varlen_fwd_step(be, r, P, step, tb, loc["q"][r], loc["k"][src], loc["v"][src], scale,
                                          lse, out, acc)
            else:
                RB.basic_varlen_fwd_step(be, r, P, step, True, tb, loc["q"][r], loc["k"][src], loc["v"][src],
                                         scale, lse, out, acc)
        outs.append(out); lses.append(lse)
        assert_close(_f(out), g.out[r], *TOL[dt]["out"], f"{g.name} out rank {r}")
        assert_close(_f(lse), g.lse[r], *TOL[dt]["out"], f"{g.name} lse rank {r}")
    st = []
    for r in range(P):
        delta = torch.empty((g.Hq, T), dtype=f32, device=dev)
        be.delta(loc["dout"][r][None], outs[r][None], delta[None])
        kshape = loc["k"][r].shape
        st.append(dict(delta=delta, dq=torch.empty((T, g.Hq, g.D), dtype=f32, device=dev),
                       dk=torch.empty(kshape, dtype=f32, device=dev), dv=torch.empty(kshape, dtype=f32, device=dev),
                       bk=torch.empty(kshape, dtype=f32, device=dev), bv=torch.empty(kshape, dtype=f32, device=dev)))
    for step in range(P):
        if step > 0:                                           # the wire: accumulators move to rank+1
            dks, dvs = [s["dk"] for s in st], [s["dv"] for s in st]
            for r in range(P):
                st[r]["dk"], st[r]["dv"] = dks[(r - 1) % P], dvs[(r - 1) % P]
        for r in range(P):
            s, src = st[r], (r - step) % P
            dst_k, dst_v = (s["dk"], s["dv"]) if step == 0 else (s["bk"], s["bv"])
            args = (loc["dout"][r], loc["q"][r], loc["k"][src], loc["v"][src], lses[r], s["delta"], scale, s["dq"],
                    dst_k, dst_v)
            if g.impl == "zigzag":
                if 0 < step <= r:
                    dst_k.zero_(); dst_v.zero_()
                RZ.zigzag_varlen_bwd_block(be, r, P, step, tb, *args)
                did = True
            else:
                did = RB.basic_varlen_bwd_block(be, r, P, step, True, tb, *args)
            if step > 0 and did:
                be.add(s["dk"], s["dk"], s["bk"]); be.add(s["dv"], s["dv"], s["bv"])
    dks, dvs = [s["dk"] for s in st], [s["dv"] for s in st]
    for r in range(P):                                         # final hop lands on the owner
        assert_close(_f(st[r]["dq"].to(tdt)), g.dq[r], *TOL[dt]["grad"], f"{g.name} dq rank {r}")
        assert_close(_f(dks[(r - 1) % P].to(tdt)), g.dk[r], *TOL[dt]["grad"], f"{g.name} dk rank {r}")
        assert_close(_f(dvs[(r - 1) % P].to(tdt)), g.dv[r], *TOL[dt]["grad"], f"{g.name} dv rank {r}")


def test_varlen_func_single_rank_autograd(dev, single_rank_pg):
    """zigzag_ring_flash_attn_varlen_func end to end at ring degree 1 (autograd glue, padded LSE)."""
    import torch.distributed as dist
    import yunchang_amd as Y
    dt, H, D = "bfloat16", 4, 128
    lens = (256, 64, 130)
    T = sum(lens)
    q, k, v, do = (_rand((T, H, D), dt, 30 + i) for i in range(4))
    tq, tk, tv = (_t(x, dt, dev).requires_grad_(True) for x in (q, k, v))
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=dev)
    out, lse, _ = Y.zigzag_ring_flash_attn_varlen_func(tq, tk, tv, cu, max(lens), causal=True,
                                                       return_attn_probs=True, group=dist.group.WORLD)
    out.backward(_t(do, dt, dev))
    ro, rl = O.varlen_attention_ref(q, k, v, np.concatenate([[0], np.cumsum(lens)]), True)
    assert_close(_f(out), ro, *TOL[dt]["out"], "out")
    assert lse.shape == (len(lens), H, max(lens))
    assert_close(_f(Y.flatten_lse(lse.detach(), cu)), rl, 2e-3, 1e-4, "lse")
    cs = np.concatenate([[0], np.cumsum(lens)])
    o16 = round_to(ro.astype(np.float32), dt)
    for i, n in enumerate(lens):
        a = cs[i]
        gq, gk, gv = O.block_bwd(do[None, a:a + n], q[None, a:a + n], k[None, a:a + n], v[None, a:a + n],
                                 o16[None, a:a + n], rl[None, :, a:a + n], None, True)
        assert_close(_f(tq.grad)[a:a + n], gq[0], *TOL[dt]["grad"], f"dq seq {i}")
        assert_close(_f(tk.grad)[a:a + n], gk[0], *TOL[dt]["grad"], f"dk seq {i}")
        assert_close(_f(tv.grad)[a:a + n], gv[0], *TOL[dt]["grad"], f"dv seq {i}")


@pytest.mark.parametrize("B,S,H,D", [(1, 130, 4, 128), (2, 200, 3, 128), (1, 333, 2, 64)])
def test_repeated_launches_on_ragged_shapes(dev, B, S, H, D):
    """Regression for an LDS reuse race that only showed on back-to-back launches of shapes with waves that own
    no valid row and a mostly out-of-range third K tile (the tile landed in Kbuf[0] while other waves were
    still reading K(0) for the first score tile): every one of 25 launches must match the oracle, forward and
    backward, and the launches must agree with each other bit for bit (the kernels are deterministic)."""
    from yunchang_amd.kernels import hip_attn_backward, hip_attn_forward
    dt = "bfloat16"
    q, k, v, do = (_rand((B, S, H, D), dt, 40 + i) for i in range(4))
    tq, tk, tv, tdo = (_t(x, dt, dev) for x in (q, k, v, do))
    ro, rl = O.block_fwd(q, k, v, None, True)
    o16 = round_to(ro.astype(np.float32), dt)
    rdq, rdk, rdv = O.block_bwd(do, q, k, v, o16, rl, None, True)
    lse_t = torch.from_numpy(rl.astype(np.float32)).to(dev)
    o16_t = _t(o16, dt, dev)
    first = None
    for rep in range(25):
        out, lse = hip_attn_forward(tq, tk, tv, 0.0, None, causal=True)
        dq, dk, dv = (torch.full_like(t, float("nan")) for t in (tq, tk, tv))
        hip_attn_backward(tdo, tq, tk, tv, o16_t, lse_t, dq, dk, dv, 0.0, None, True)
        got = [_f(x) for x in (out, lse, dq, dk, dv)]
        if first is None:
            first = got
            assert_close(got[0], ro, *TOL[dt]["out"], "out")
            assert_close(got[1], rl, 2e-3, 1e-4, "lse")
            for g_, r_, n_ in zip(got[2:], (rdq, rdk, rdv), ("dq", "dk", "dv")):
                assert_close(g_, r_, *TOL[dt]["grad"], n_)
        else:
            for a_, b_, n_ in zip(got, first, ("out", "lse", "dq", "dk", "dv")):
                assert np.array_equal(a_, b_), f"launch {rep} differs from launch 0 in {n_}"


def test_c2_full_size_vs_the_reference_third_party_op(dev):
    """BASELINE configs[1] at full size (B2 S8192 H16 D128 bf16 causal) against the op the reference's
    TORCH_EFFICIENT path runs on this GPU: aten::_scaled_dot_product_efficient_attention on (B,H,S,D) views
    (yunchang/kernels/attention.py:76-86) -- the literal parity target of SURVEY.md 8(c).  Both results are
    bf16-rounded, so the bound is two bf16 roundings of |out| <= ~4 plus accumulation-order noise."""
    from yunchang_amd.kernels import hip_attn_forward
    B, S, H, D = 2, 8192, 16, 128
    g = torch.Generator(device=dev).manual_seed(7)
    q, k, v = (torch.randn((B, S, H, D), device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
    try:
        ref_out, ref_lse = torch.ops.aten._scaled_dot_product_efficient_attention(
            q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), None, True, 0.0, True, scale=D ** -0.5)[:2]
    except Exception as e:                     # pragma: no cover - depends on the torch build
        pytest.skip(f"aten efficient attention does not run here: {e!r}")
    out, lse = hip_attn_forward(q, k, v, 0.0, None, causal=True)
    err = (out.float() - ref_out.transpose(1, 2).float()).abs()
    lim = 2e-2 + 2e-2 * ref_out.transpose(1, 2).float().abs()
    assert bool((err <= lim).all()), f"max abs err {float(err.max()):.3e}"
    assert float((lse - ref_lse.float()).abs().max()) < 2e-3


def test_c2_full_size_backward_vs_torch_sdpa(dev):
    """BASELINE configs[1] at full size, forward + backward, against torch's own scaled_dot_product_attention
    autograd on this GPU (the kernels a ROCm PyTorch ships; the reference's TORCH path has no backward at all,
    kernels/attention.py:138-159).  Stated gradient tolerance (golden_util.TOL)."""
    import torch.nn.functional as F
    from yunchang_amd.kernels import hip_attn_func
    B, S, H, D = 2, 8192, 16, 128
    g = torch.Generator(device=dev).manual_seed(11)
    q, k, v, do = (torch.randn((B, S, H, D), device=dev, generator=g).to(torch.bfloat16) for _ in range(4))
    ours = [t.clone().requires_grad_(True) for t in (q, k, v)]
    hip_attn_func(*ours, causal=True).backward(do)
    ref = [t.transpose(1, 2).clone().requires_grad_(True) for t in (q, k, v)]
    try:
        F.scaled_dot_product_attention(*ref, is_causal=True).backward(do.transpose(1, 2))
    except Exception as e:                     # pragma: no cover - depends on the torch build
        pytest.skip(f"torch SDPA backward does not run here: {e!r}")
    atol, rtol = TOL["bfloat16"]["grad"]
    for a, b, name in zip(ours, ref, ("dq", "dk", "dv")):
        ga, gb = a.grad.float(), b.grad.transpose(1, 2).float()
        err = (ga - gb).abs()
        assert bool((err <= atol + rtol * gb.abs()).all()), f"{name}: max abs err {float(err.max()):.3e}"




@pytest.mark.parametrize("B,Sq,Sk,Hq,Hkv,D,causal,dt", [(1, 1024, 1024, 2, 2, 128, True, "bfloat16"),
                                                        (2, 300, 712, 4, 2, 64, False, "float16"),
                                                        (1, 333, 200, 2, 1, 128, True, "bfloat16"),
                                                        (1, 4096, 4096, 2, 1, 128, True, "bfloat16")])
def test_forward_k_split_through_the_binding(dev, B, Sq, Sk, Hq, Hkv, D, causal, dt):
    """usp_fwd_args.k_splits through _C.flash_fwd (workspace from torch): every n against the unsplit launch (fp32
    summation order is all that differs) and the unsplit launch against the oracle; plain, partly final, merged."""
    from yunchang_amd import _C
    q, k, v = (_rand(s, dt, 50 + i) for i, s in enumerate([(B, Sq, Hq, D), (B, Sk, Hkv, D), (B, Sk, Hkv, D)]))
    tq, tk, tv = (_t(x, dt, dev) for x in (q, k, v))
    scale = D ** -0.5

    def run(n, final_end, merged):
        out = torch.full((B, Sq, Hq, D), float("nan"), dtype=getattr(torch, dt), device=dev)
        acc = torch.full((B, Sq, Hq, D), float("nan"), dtype=torch.float32, device=dev)
        lse = torch.full((B, Hq, Sq), float("nan"), dtype=torch.float32, device=dev)
        if merged:       # keys [0,h) first into acc (unsplit), then the rest merged in by the launch under test
            h = (Sk // 2) & ~7
            _C.flash_fwd(tq, tk[:, :h], tv[:, :h], scale, False, lse, out, acc, False, 0, 0, k_splits=0)
            _C.flash_fwd(tq, tk[:, h:], tv[:, h:], scale, False, lse, out, acc, True, 0, final_end, k_splits=n)
        else:
            _C.flash_fwd(tq, tk, tv, scale, causal, lse, out, acc, False, 0, final_end, k_splits=n)
        torch.cuda.synchronize()
        res = torch.where((torch.arange(Sq, device=dev) < final_end)[None, :, None, None], out.float(), acc)
        return res.cpu().numpy(), lse.cpu().numpy()

    ro, rl = O.attention_ref(q, k, v, causal=causal)
    for final_end, merged in ((Sq, False), (Sq // 2, False)) + (((Sq, True),) if not causal else ()):
        base_o, base_l = run(0, final_end, merged)
        if not merged:
            assert_close(base_o, ro, *TOL[dt]["out"], f"unsplit fe={final_end}")
        for n in (2, 3, 4, 8):
            o, l = run(n, final_end, merged)
            assert np.isfinite(o).all() and not np.isnan(l).any()
            assert_close(o, base_o, *TOL[dt]["out"], f"k_splits={n} fe={final_end} merged={merged}")
            assert_close(l, base_l, 1e-5, 1e-5, f"lse k_splits={n}")


def test_expanded_gradient_reaches_the_kernels(dev, single_rank_pg):
    """`out.sum().backward()` hands the backward an expanded scalar (every stride 0) and a caller may hold q/k/v views
    with a non-unit head-dim stride: both are normalised in front of the kernels (round 2; found by the test backend's
    operand checks).  Gradients against the oracle."""
    import yunchang_amd as Y
    B, S, H, D = 1, 384, 2, 64
    q, k, v = (_rand((B, S, H, D), "bfloat16", 70 + i) for i in range(3))
    wide = []
    for x in (q, k, v):
        w = torch.zeros((B, S, H, 2 * D), dtype=torch.bfloat16, device=dev)
        w[..., ::2] = _t(x, "bfloat16", dev)
        wide.append(w.requires_grad_(True))
    out = Y.zigzag_ring_flash_attn_func(*(w[..., ::2] for w in wide), causal=True, group=Y.PROCESS_GROUP.RING_PG)
    out.float().sum().backward()
    ro, rl = O.attention_ref(q, k, v, causal=True)
    assert_close(_f(out), ro, *TOL["bfloat16"]["out"], "out")
    rdq, rdk, rdv = O.block_bwd(np.ones_like(ro, dtype=np.float32), q, k, v, ro, rl, None, True)
    for w, ref, name in zip(wide, (rdq, rdk, rdv), ("dq", "dk", "dv")):
        assert_close(_f(w.grad[..., ::2]), ref, *TOL["bfloat16"]["grad"], name)
        assert (w.grad[..., 1::2] == 0).all()


@pytest.mark.parametrize("Hq,Hkv", [(16, 2), (16, 16)], ids=["configs4_rank_gqa", "configs3_rank_mha"])
def test_c5_rank_block_shapes_against_sampled_fp64(dev, Hq, Hkv):
    """The blocks a rank of BASELINE configs[4] (8 GPUs, ulysses 2 x ring 4, S = 65536, H32/Hkv4) actually launches, at
    their REAL size: c = 8192, local q (1, 16384, 16, 128), K/V (1, 16384, 2, 128) -- and the same with 16 KV heads, the
    blocks of a configs[3] rank (4 GPUs, ring 4, S = 32768, MHA H16; c = 4096 there, the larger c is the harder case).  Forward: step 0 (causal, all rows,
    nothing final) then a step beyond the rank (q[c:] x another rank's 16384 keys, merge mode, final_end = c) -- rows
    [c, 2c) final in 16 bits, rows [0, c) still fp32 in the running output.  Backward: the block of that step (global
    LSE, delta, dq accumulated onto a running fp32 buffer, fp32 dK/dV through the GQA head-split workspace).  Sampled
    rows / key columns against exact fp64 attention over the keys those rows have seen."""
    from yunchang_amd import _C
    torch.manual_seed(5)
    c, D = 8192, 128
    S2, G, scale = 2 * c, Hq // Hkv, 128 ** -0.5
    q, do = (torch.randn(1, S2, Hq, D, device=dev).to(torch.bfloat16) for _ in range(2))
    kl, vl, ko, vo = (torch.randn(1, S2, Hkv, D, device=dev).to(torch.bfloat16) for _ in range(4))
    out = torch.full((1, S2, Hq, D), float("nan"), dtype=torch.bfloat16, device=dev)
    acc = torch.full((1, S2, Hq, D), float("nan"), dtype=torch.float32, device=dev)
    lse = torch.full((1, Hq, S2), float("nan"), dtype=torch.float32, device=dev)
    _C.flash_fwd(q, kl, vl, scale, True, lse, out, acc, False, 0, 0, interleave=True)                  # step 0
    _C.flash_fwd(q[:, c:], ko, vo, scale, False, lse[:, :, c:], out[:, c:], acc[:, c:], True, 0, c,    # step > rank
                 interleave=True)
    rs = np.random.RandomState(0)
    heads = (0, 9, 15)
    for h in heads:
        kld, vld, kod, vod = (t[0, :, h // G].double() for t in (kl, vl, ko, vo))
        for i in sorted({c, S2 - 1, *rs.randint(c, S2, 4).tolist()}):       # rows that saw both blocks: final, 16-bit
            s = torch.cat([kod @ q[0, i, h].double(), kld[:i + 1] @ q[0, i, h].double()]) * scale
            l = torch.logsumexp(s, 0)
            ref = torch.exp(s - l) @ torch.cat([vod, vld[:i + 1]])
            assert_close(_f(out[0, i, h]), ref.cpu().numpy(), *TOL["bfloat16"]["out"], f"final row {i} head {h}")
            assert abs(float(lse[0, h, i]) - float(l)) < 2e-3
        for i in sorted({0, c - 1, *rs.randint(0, c, 3).tolist()}):         # rows of the front half: fp32, not final
            s = (kld[:i + 1] @ q[0, i, h].double()) * scale
            l = torch.logsumexp(s, 0)
            ref = torch.exp(s - l) @ vld[:i + 1]
            assert_close(_f(acc[0, i, h]), ref.cpu().numpy(), *TOL["bfloat16"]["out"], f"running row {i} head {h}")
            assert abs(float(lse[0, h, i]) - float(l)) < 2e-3
    assert torch.isnan(out[:, :c].float()).all()                             # nothing emitted for rows that are not final
    # ---- backward block of the same step: dout / q / lse / delta rows [c, 2c) against the other rank's K/V ----------
    delta = torch.empty((1, Hq, S2), dtype=torch.float32, device=dev)
    _C.bwd_delta(do[:, c:], out[:, c:], delta[:, :, c:])
    dq_acc = torch.ones((1, S2, Hq, D), dtype=torch.float32, device=dev)     # a running buffer: the block is ADDED
    dk = torch.full((1, S2, Hkv, D), float("nan"), dtype=torch.float32, device=dev)
    dv = torch.full_like(dk, float("nan"))
    _C.flash_bwd(do[:, c:], q[:, c:], ko, vo, lse[:, :, c:], delta[:, :, c:], dq_acc[:, c:], dk, dv, scale, False,
                 accum_dq=True, interleave=True)
    assert (dq_acc[:, :c] == 1).all()
    atol, rtol = grad_tol("bfloat16", G)
    for h in heads:
        kod, vod = ko[0, :, h // G].double(), vo[0, :, h // G].double()
        for i in sorted({c, S2 - 1, *rs.randint(c, S2, 3).tolist()}):
            qi, doi = q[0, i, h].double(), do[0, i, h].double()
            p = torch.exp((kod @ qi) * scale - lse[0, h, i].double())
            ds = p * (vod @ doi - delta[0, h, i].double())
            assert_close(_f(dq_acc[0, i, h]) - 1.0, ((ds @ kod) * scale).cpu().numpy(), *TOL["bfloat16"]["grad"],
                         f"dq row {i} head {h}")
    for hk in (0, Hkv - 1):
        kod, vod = ko[0, :, hk].double(), vo[0, :, hk].double()
        for j in sorted({0, S2 - 1, *rs.randint(0, S2, 3).tolist()}):
            rdk = torch.zeros(D, dtype=torch.float64, device=dev)
            rdv = torch.zeros_like(rdk)
            for g in range(G):
                h = hk * G + g
                qi, doi = q[0, c:, h].double(), do[0, c:, h].double()
                p = torch.exp((qi @ kod[j]) * scale - lse[0, h, c:].double())
                rdv += p @ doi
                rdk += ((p * (doi @ vod[j] - delta[0, h, c:].double())) @ qi) * scale
            assert_close(_f(dk[0, j, hk]), rdk.cpu().numpy(), atol, rtol, f"dk key {j} kv head {hk}")
            assert_close(_f(dv[0, j, hk]), rdv.cpu().numpy(), atol, rtol, f"dv key {j} kv head {hk}")


def _load_bench():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def test_seq64k_sampled_parity_through_bench(dev):
    """The metric's own size on one GPU (B1 S65536 H32/Hkv4 D128 causal, forward + backward kernels) against exact fp64
    attention on sampled rows and key columns -- the check bench.py attaches to `roofline.sampled_parity` (round 5: the N = 1
    workload itself)."""
    b = _load_bench()
    c5 = b.WORKLOADS[8]
    t = b._fwd_bwd_kernels(c5["B"], c5["S"], c5["Hq"], c5["Hkv"], c5["D"], dev, 1, keep=True)
    par = b.sampled_parity(t["tensors"])
    err, ratio = par["max_abs_err"], par["max_err_over_tolerance"]
    # every gate is `x < bound`: False for NaN (bench.nanmax propagates a NaN of any sampled row into the figure).
    # out / lse / dq: pure absolute bounds.  dk / dv are sums over 8 query heads x up to 65536 rows with entries well above 1:
    # gated by the comparator of every other parity test, |err| <= atol + rtol |want| with the stated bf16 gradient
    # tolerance (5e-2, 5e-2), un-widened for the group size.  (Measured: dk 1.7e-2, dv 5.3e-2 absolute = 0.3 / 0.6 of the
    # tolerance; the dK/dV kernel rounds K * scale * log2(e) to bf16 once per item, which adds ~1e-3 relative to P.)
    assert err["out"] < 2e-2 and err["lse"] < 2e-3 and err["dq"] < 5e-2, par
    assert ratio["dk"] < 1.0 and ratio["dv"] < 1.0 and ratio["dq"] < 1.0 and ratio["out"] < 1.0, par


@pytest.mark.parametrize("B,Sq,Sk,Hq,Hkv,D,causal,dt", [(1, 1024, 1024, 2, 2, 128, True, "bfloat16"),
                                                        (2, 300, 712, 4, 2, 64, False, "float16"),
                                                        (1, 333, 200, 2, 1, 128, True, "bfloat16"),
                                                        (1, 4096, 4096, 2, 1, 128, True, "bfloat16")])
def test_backward_cuts_through_the_binding(dev, B, Sq, Sk, Hq, Hkv, D, causal, dt):
    """usp_bwd_args.dq_splits / dkdv_splits (ABI v5) through _C.flash_bwd (workspace from torch): every cut against the
    uncut launch (fp32 summation order is all that differs), the uncut launch against the oracle; fp32 outputs with an
    accumulated dq, and 16-bit final outputs.  The default policy must pick cuts for the few-head long shape."""
    from yunchang_amd import _C
    q, k, v, do = (_rand(s, dt, 80 + i) for i, s in enumerate([(B, Sq, Hq, D), (B, Sk, Hkv, D), (B, Sk, Hkv, D), (B, Sq, Hq, D)]))
    tq, tk, tv, tdo = (_t(x, dt, dev) for x in (q, k, v, do))
    scale = D ** -0.5
    out = torch.empty_like(tq)
    lse = torch.empty((B, Hq, Sq), dtype=torch.float32, device=dev)
    _C.flash_fwd(tq, tk, tv, scale, causal, lse, out)
    delta = torch.empty_like(lse)
    _C.bwd_delta(tdo, out, delta)
    ro, rl = O.attention_ref(q, k, v, causal=causal)
    refs = O.block_bwd(do, q, k, v, ro, rl, None, causal)
    tdt = getattr(torch, dt)

    def run(splits, final16):
        if final16:
            g = [torch.full(t.shape, float("nan"), dtype=tdt, device=dev) for t in (tq, tk, tv)]
            _C.flash_bwd(tdo, tq, tk, tv, lse, delta, None, None, None, scale, causal, dq16=g[0], dk16=g[1], dv16=g[2],
                         splits=splits)
            return [_f(t) for t in g]
        dq = torch.ones((B, Sq, Hq, D), dtype=torch.float32, device=dev)            # accumulated onto
        dk = torch.full((B, Sk, Hkv, D), float("nan"), dtype=torch.float32, device=dev)
        dv = torch.full_like(dk, float("nan"))
        _C.flash_bwd(tdo, tq, tk, tv, lse, delta, dq, dk, dv, scale, causal, accum_dq=True, splits=splits)
        return [_f(dq) - 1.0, _f(dk), _f(dv)]

    atol, rtol = grad_tol(dt, Hq // Hkv)
    for final16 in (False, True):
        base = run((0, 0), final16)
        for got, ref, name in zip(base, refs, ("dq", "dk", "dv")):
            assert_close(got, ref, atol, rtol, f"uncut {name} final16={final16}")
        for splits in ((2, 2), (3, 4), (8, 8), (4, 1), (1, 3)):
            for got, b0, name in zip(run(splits, final16), base, ("dq", "dk", "dv")):
                assert np.isfinite(got).all(), f"{name} cuts {splits}"
                assert_close(got, b0, atol, rtol, f"{name} cuts {splits} final16={final16}")
    if Sq == 4096:
        assert _C.bwd_splits(B, Sq, Sk, Hq, causal) == (4, 4)        # 32 / 64 items: the policy cuts this launch
        for got, ref, name in zip(run(None, True), refs, ("dq", "dk", "dv")):
            assert_close(got, ref, atol, rtol, f"default policy {name}")


WINDOWS = [
    # B, Sq, Sk, Hq, Hkv, D, causal, (left, right), dtype
    (1, 512, 512, 2, 2, 128, True, (64, 0), "bfloat16"),        # causal sliding window (the usual local attention)
    (2, 384, 640, 4, 2, 64, False, (100, 30), "bfloat16"),      # both bounds, Sq != Sk, GQA
    (1, 300, 300, 2, 1, 128, False, (-1, 17), "float16"),       # right bound only = a shifted causal limit
    (1, 333, 200, 2, 2, 64, False, (33, -1), "bfloat16"),       # left bound only; rows without any visible key
    (1, 2048, 2048, 2, 2, 128, True, (256, 0), "bfloat16"),     # long: whole key tiles left of the window are skipped
    (1, 1024, 2048, 2, 1, 32, True, (0, 0), "bfloat16"),        # the diagonal only
    (1, 777, 777, 3, 3, 128, False, (5000, 5000), "bfloat16"),  # a window wider than the sequence = full attention
]


@pytest.mark.parametrize("B,Sq,Sk,Hq,Hkv,D,causal,win,dt", WINDOWS)
def test_sliding_window_forward_backward_vs_oracle(dev, B, Sq, Sk, Hq, Hkv, D, causal, win, dt):
    """flash-attn's `window_size` (left, right) through the C ABI (USP_ATTN_WINDOW, ABI v5) -- the argument the
    reference's block contract passes through (kernels/attention.py:165-202) -- forward (out, LSE) and backward against the
    fp64 oracle, whose window mask is pinned to the reference's (tests/golden/w_window_ref.npz); also with the K split
    of the forward and the cuts of the backward on top."""
    from yunchang_amd import _C
    q, k, v, do = (_rand(s, dt, 90 + i) for i, s in enumerate([(B, Sq, Hq, D), (B, Sk, Hkv, D), (B, Sk, Hkv, D), (B, Sq, Hq, D)]))
    tq, tk, tv, tdo = (_t(x, dt, dev) for x in (q, k, v, do))
    scale = D ** -0.5
    ro, rl = O.attention_ref(q, k, v, causal=causal, window=win)
    rdq, rdk, rdv = O.block_bwd(do, q, k, v, ro, rl, None, causal, window=win)
    tdt = getattr(torch, dt)
    for ks, cuts in ((0, (0, 0)), (3, (2, 3))):
        out = torch.full((B, Sq, Hq, D), float("nan"), dtype=tdt, device=dev)
        lse = torch.full((B, Hq, Sq), float("nan"), dtype=torch.float32, device=dev)
        _C.flash_fwd(tq, tk, tv, scale, causal, lse, out, window=win, k_splits=ks)
        assert_close(_f(out), ro, *TOL[dt]["out"], f"out k_splits={ks}")
        lg, seen = _f(lse), np.isfinite(rl)
        assert (np.isneginf(lg) == ~seen).all(), "rows without a visible key must report lse = -inf"
        assert_close(lg[seen], rl[seen], 2e-3, 1e-4, f"lse k_splits={ks}")
        delta = torch.empty_like(lse)
        _C.bwd_delta(tdo, out, delta)
        g = [torch.full(t.shape, float("nan"), dtype=tdt, device=dev) for t in (tq, tk, tv)]
        _C.flash_bwd(tdo, tq, tk, tv, lse, delta, None, None, None, scale, causal, dq16=g[0], dk16=g[1], dv16=g[2],
                     window=win, splits=cuts)
        atol, rtol = grad_tol(dt, Hq // Hkv)
        for got, ref, name in zip(g, (rdq, rdk, rdv), ("dq", "dk", "dv")):
            assert_close(_f(got), ref, atol, rtol, f"{name} cuts={cuts}")


def test_sliding_window_through_the_block_contract_and_the_layer(dev, single_rank_pg):
    """window_size through the three seams the reference's selector hands out (fwd-only / bwd-only / fwd-bwd callables)
    and through LongContextAttention on a 1-rank grid (ring degree 1: one block)."""
    import yunchang_amd as Y
    from yunchang_amd.kernels import select_flash_attn_impl
    B, S, H, D, win = 1, 640, 2, 64, (96, 0)
    q, k, v, do = (_rand((B, S, H, D), "bfloat16", 110 + i) for i in range(4))
    ro, rl = O.attention_ref(q, k, v, causal=True, window=win)
    refs = O.block_bwd(do, q, k, v, ro, rl, None, True, window=win)
    tq, tk, tv, tdo = (_t(x, "bfloat16", dev) for x in (q, k, v, do))
    fwd = select_flash_attn_impl(Y.AttnType.HIP, stage="fwd-only")
    out, lse = fwd(tq, tk, tv, 0.0, None, causal=True, window_size=win)
    assert_close(_f(out), ro, *TOL["bfloat16"]["out"], "fwd-only")
    bwd = select_flash_attn_impl(Y.AttnType.HIP, stage="bwd-only")
    g = [torch.empty_like(t) for t in (tq, tk, tv)]
    bwd(tdo, tq, tk, tv, out, lse, g[0], g[1], g[2], 0.0, None, True, win, 0.0, None, False)
    for got, ref, name in zip(g, refs, ("dq", "dk", "dv")):
        assert_close(_f(got), ref, *TOL["bfloat16"]["grad"], f"bwd-only {name}")
    leaves = [t.clone().requires_grad_(True) for t in (tq, tk, tv)]
    attn = Y.LongContextAttention(ring_impl_type="basic", attn_type=Y.AttnType.HIP)
    o2 = attn(*leaves, causal=True, window_size=win)
    o2.backward(tdo)
    assert_close(_f(o2), ro, *TOL["bfloat16"]["out"], "layer out")
    for leaf, ref, name in zip(leaves, refs, ("dq", "dk", "dv")):
        assert_close(_f(leaf.grad), ref, *TOL["bfloat16"]["grad"], f"layer {name}")


@pytest.mark.parametrize("D", [96, 80, 40, 120])
def test_head_dims_between_the_instantiated_ones(dev, single_rank_pg, D):
    """Head dims flash-attn serves and this library does not instantiate (multiples of 8 below 128): the block contract
    (fwd-only / bwd-only / fwd-bwd) and the layer run them on zero-padded copies, against the oracle at the ORIGINAL head
    dim; above 128 the refusal is a NotImplementedError, not a wrong answer."""
    import yunchang_amd as Y
    from yunchang_amd.kernels import hip_attn_backward, hip_attn_forward, hip_attn_func
    B, S, Hq, Hkv = 1, 320, 4, 2
    q, k, v, do = (_rand(s, "bfloat16", 120 + i) for i, s in enumerate([(B, S, Hq, D), (B, S, Hkv, D), (B, S, Hkv, D), (B, S, Hq, D)]))
    ro, rl = O.attention_ref(q, k, v, causal=True)
    refs = O.block_bwd(do, q, k, v, ro, rl, None, True)
    tq, tk, tv, tdo = (_t(x, "bfloat16", dev) for x in (q, k, v, do))
    out, lse = hip_attn_forward(tq, tk, tv, causal=True)
    assert out.shape == tq.shape
    assert_close(_f(out), ro, *TOL["bfloat16"]["out"], "fwd-only")
    assert_close(_f(lse), rl, 2e-3, 1e-4, "lse")
    g = [torch.empty_like(t) for t in (tq, tk, tv)]
    hip_attn_backward(tdo, tq, tk, tv, out, lse, g[0], g[1], g[2], 0.0, None, True)
    atol, rtol = grad_tol("bfloat16", Hq // Hkv)
    for got, ref, name in zip(g, refs, ("dq", "dk", "dv")):
        assert_close(_f(got), ref, atol, rtol, f"bwd-only {name}")
    for fn in (lambda a, b, c: hip_attn_func(a, b, c, causal=True),
               lambda a, b, c: Y.LongContextAttention(ring_impl_type="zigzag", attn_type=Y.AttnType.HIP)(a, b, c, causal=True)):
        leaves = [t.clone().requires_grad_(True) for t in (tq, tk, tv)]
        o2 = fn(*leaves)
        o2.backward(tdo)
        assert_close(_f(o2), ro, *TOL["bfloat16"]["out"], "fwd-bwd out")
        for leaf, ref, name in zip(leaves, refs, ("dq", "dk", "dv")):
            assert_close(_f(leaf.grad), ref, atol, rtol, f"fwd-bwd {name}")
    big = torch.zeros((1, 64, 2, 256), dtype=torch.bfloat16, device=dev)
    with pytest.raises(NotImplementedError):
        hip_attn_forward(big, big, big, causal=True)


def test_kernel_rate_probe_measures_a_plausible_rate(dev):
    """comm/link.py:_probe_kernel_rate (run beside the link probe at set_seq_parallel_pg when USP_LINK_PROBE=1): the forward
    kernel on a part-filling launch, between 0.2 and 2.5 PFLOP/s on an MI355X."""
    import yunchang_amd.comm.link as L
    rate = L._probe_kernel_rate(dev)
    assert 2e14 < rate < 2.5e15, rate
