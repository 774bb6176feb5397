"""Randomised GPU parity sweep: many small ragged shapes (dense and packed, MHA and GQA, causal and full,
Sq != Sk, every head dim, both dtypes) through the C ABI against the CPU oracle, every call launched twice and
required to be bit-identical (the kernels are deterministic; a difference means a race).  Sizes keep the fp64
oracle at a few milliseconds per case.  Seeds are fixed: failures reproduce."""
import numpy as np
import pytest
import torch

from golden_util import TOL, assert_close, round_to
from oracle import usp_oracle as O

pytestmark = pytest.mark.gpu

import os
_N_DENSE = int(os.environ.get("USP_FUZZ_DENSE", "28"))      # larger sweeps: USP_FUZZ_DENSE=400 USP_FUZZ_PACKED=200
_N_PACKED = int(os.environ.get("USP_FUZZ_PACKED", "14"))


@pytest.fixture(scope="module")
def dev():
    from yunchang_amd import _C
    _C.load()
    return torch.device("cuda:0")


def _t(x, dt, dev):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(getattr(torch, dt)).to(dev)


def _f(t):
    return t.detach().float().cpu().numpy()


def _dense_case(rs):
    D = int(rs.choice([32, 64, 128]))
    dt = str(rs.choice(["bfloat16", "float16"]))
    Hkv = int(rs.choice([1, 2, 3]))
    Hq = Hkv * int(rs.choice([1, 2, 4]))
    B = int(rs.choice([1, 2, 3]))
    Sq = int(rs.randint(1, 700))
    Sk = Sq if rs.rand() < 0.5 else int(rs.randint(1, 700))
    causal = bool(rs.rand() < 0.6)
    # (drawn last, so that the shapes of the earlier rounds' seeds stay what they were)  a third of the cases carry a
    # sliding window, a third the K split of the forward and cuts of the two backward launches (ABI v5)
    win = None
    if rs.rand() < 0.33:
        win = (int(rs.choice([-1, 0, 1, 17, 64, 200, 1000])), int(rs.choice([-1, 0, 3, 40])))
        if win == (-1, -1):
            win = (33, 0)
    ks, cuts = 0, (0, 0)
    if rs.rand() < 0.33:
        ks, cuts = int(rs.choice([2, 3, 5, 8])), (int(rs.choice([0, 2, 3, 8])), int(rs.choice([0, 2, 4])))
    return B, Sq, Sk, Hq, Hkv, D, causal, dt, win, ks, cuts


@pytest.mark.parametrize("seed", range(_N_DENSE))
def test_fuzz_dense(dev, seed):
    rs = np.random.RandomState(1000 + seed)
    _run_dense(dev, rs, _dense_case(rs))


_N_ROW64 = int(os.environ.get("USP_FUZZ_ROW64", "16"))      # larger sweeps: USP_FUZZ_ROW64=300


def _row64_case(rs):
    """Shapes the one-wave-per-SIMD BACKWARD kernels serve by the library's own dispatch (D = 128, dense, no window, no
    cuts; both dtypes): they take every such launch, however small.  The 4 x 64 FORWARD is dispatched from 256 work items
    up, which no shape here has -- this sweep's forward runs on the 4-wave kernel; the forced sweep of
    tests/test_gpu_row64.py is the one that puts ragged shapes through `flash_fwd64_kernel`.  Longer sequences than the
    general sweep: several 256-row query blocks and 128-key blocks per head, ragged ends, Sq != Sk in both directions,
    GQA groups up to 8."""
    dt = str(rs.choice(["bfloat16", "bfloat16", "float16"]))
    Hkv = int(rs.choice([1, 2]))
    Hq = Hkv * int(rs.choice([1, 2, 4, 8]))
    B = int(rs.choice([1, 2]))
    Sq = int(rs.choice([rs.randint(1, 130), rs.randint(130, 1400)]))
    Sk = Sq if rs.rand() < 0.5 else int(rs.choice([rs.randint(1, 130), rs.randint(130, 1400)]))
    causal = bool(rs.rand() < 0.6)
    return B, Sq, Sk, Hq, Hkv, 128, causal, dt, None, 0, (0, 0)


@pytest.mark.parametrize("seed", range(_N_ROW64))
def test_fuzz_64row_kernels(dev, seed):
    rs = np.random.RandomState(5000 + seed)
    _run_dense(dev, rs, _row64_case(rs))


def _run_dense(dev, rs, case):
    from yunchang_amd import _C
    B, Sq, Sk, Hq, Hkv, D, causal, dt, win, ks, cuts = case
    what = f"B{B} Sq{Sq} Sk{Sk} Hq{Hq} Hkv{Hkv} D{D} causal={causal} {dt} window={win} k_splits={ks} cuts={cuts}"
    wkw = {} if win is None else {"window": win}
    q, k, v, do = (round_to(rs.standard_normal(s).astype(np.float32), dt)
                   for s in [(B, Sq, Hq, D), (B, Sk, Hkv, D), (B, Sk, Hkv, D), (B, Sq, Hq, D)])
    tq, tk, tv, tdo = (_t(x, dt, dev) for x in (q, k, v, do))
    scale = D ** -0.5
    ro, rl = O.attention_ref(q, k, v, causal, scale, **wkw)
    runs = []
    for _ in range(2):
        out = torch.full((B, Sq, Hq, D), float("nan"), dtype=tq.dtype, device=dev)
        lse = torch.full((B, Hq, Sq), float("nan"), dtype=torch.float32, device=dev)
        _C.flash_fwd(tq, tk, tv, scale, causal, lse, out=out, interleave=len(runs) == 1, k_splits=ks, **wkw)
        runs.append((_f(out), _f(lse)))
    assert np.array_equal(runs[0][0], runs[1][0], equal_nan=True) and np.array_equal(runs[0][1], runs[1][1]), what
    fin = np.isfinite(rl)
    assert (np.isfinite(runs[0][1]) == fin).all(), what + ": empty rows must give lse = -inf"
    assert_close(runs[0][0], ro, *TOL[dt]["out"], what + " out")
    assert_close(runs[0][1][fin], rl[fin], 2e-3, 1e-4, what + " lse")
    o16 = round_to(ro.astype(np.float32), dt)
    rdq, rdk, rdv = O.block_bwd(do, q, k, v, o16, rl, scale, causal, **wkw)
    lse_t = torch.from_numpy(np.ascontiguousarray(rl, dtype=np.float32)).to(dev)
    delta = torch.empty((B, Hq, Sq), dtype=torch.float32, device=dev)
    _C.bwd_delta(tdo, _t(o16, dt, dev), delta)
    grads = []
    for _ in range(2):
        dq, dk, dv = (torch.full_like(t, float("nan")) for t in (tq, tk, tv))
        _C.flash_bwd(tdo, tq, tk, tv, lse_t, delta, None, None, None, scale, causal, dq16=dq, dk16=dk, dv16=dv,
                     interleave=len(grads) == 1, splits=cuts, **wkw)
        grads.append([_f(x) for x in (dq, dk, dv)])
    for a_, b_, n_ in zip(grads[0], grads[1], ("dq", "dk", "dv")):
        assert np.array_equal(a_, b_), f"{what}: {n_} differs between two launches"
    for g_, r_, n_ in zip(grads[0], (rdq, rdk, rdv), ("dq", "dk", "dv")):
        # A gradient entry is a sum of N products whose 16-bit factors (P, dS) carry a relative rounding error of
        # 2^-9 / sqrt(3) each: the absolute error of the SUM is about 1.1e-3 x rms(entry) per sigma, whatever the entry's own
        # value -- an entry near a zero crossing of a tensor whose entries are ~20 (1000+ rows per key, ten keys: P is not
        # small) misses `atol + rtol |want|` by 2x in the 8-wave AND the 64-row kernels alike.  So for LONG sums only
        # (>= 1000 products per entry: dK / dV sum Sq * G rows, dQ sums Sk keys) the absolute part of the bound does not drop
        # below ~7 sigma of that noise; every other case keeps the stated tolerance.  Pinned by
        # tests/test_gpu_row64.py::test_long_sum_gradient_noise_is_that_of_16bit_products (both families miss the
        # un-floored bound by the same amount on that case and pass it against 16-bit-rounded products).
        atol, rtol = TOL[dt]["grad"]
        n_sum = Sk if n_ == "dq" else Sq * (Hq // Hkv)
        if n_sum >= 1000:
            atol = max(atol, 8e-3 * float(np.sqrt(np.mean(np.square(r_, dtype=np.float64)))))
        assert_close(g_, r_, atol, rtol, f"{what} {n_}")


@pytest.mark.parametrize("seed", range(_N_PACKED))
def test_fuzz_packed(dev, seed):
    from yunchang_amd import _C
    rs = np.random.RandomState(2000 + seed)
    D = int(rs.choice([32, 64, 128]))
    dt = str(rs.choice(["bfloat16", "float16"]))
    Hkv = int(rs.choice([1, 2]))
    Hq = Hkv * int(rs.choice([1, 2, 4]))
    n = int(rs.randint(1, 7))
    lq = [int(x) for x in rs.randint(1, 400, size=n)]
    same = rs.rand() < 0.6
    lk = lq if same else [int(x) for x in rs.randint(1, 400, size=n)]
    causal = bool(rs.rand() < 0.7)
    what = f"lens_q={lq} lens_k={lk} Hq{Hq} Hkv{Hkv} D{D} causal={causal} {dt}"
    Tq, Tk = sum(lq), sum(lk)
    q, k, v, do = (round_to(rs.standard_normal(s).astype(np.float32), dt)
                   for s in [(Tq, Hq, D), (Tk, Hkv, D), (Tk, Hkv, D), (Tq, Hq, D)])
    tq, tk, tv, tdo = (_t(x, dt, dev) for x in (q, k, v, do))
    cq, ck = np.concatenate([[0], np.cumsum(lq)]), np.concatenate([[0], np.cumsum(lk)])
    sq = torch.tensor(np.stack([cq[:-1], lq], 1), dtype=torch.int32, device=dev)
    sk = torch.tensor(np.stack([ck[:-1], lk], 1), dtype=torch.int32, device=dev)
    scale = D ** -0.5
    ro = np.zeros((Tq, Hq, D)); rl = np.zeros((Hq, Tq))
    rdq = np.zeros((Tq, Hq, D)); rdk = np.zeros((Tk, Hkv, D)); rdv = np.zeros((Tk, Hkv, D))
    for i in range(n):
        a, b, c, d = cq[i], cq[i + 1], ck[i], ck[i + 1]
        o_i, l_i = O.block_fwd(q[None, a:b], k[None, c:d], v[None, c:d], scale, causal)
        ro[a:b], rl[:, a:b] = o_i[0], l_i[0]
    o16 = round_to(ro.astype(np.float32), dt)
    for i in range(n):
        a, b, c, d = cq[i], cq[i + 1], ck[i], ck[i + 1]
        g = O.block_bwd(do[None, a:b], q[None, a:b], k[None, c:d], v[None, c:d], o16[None, a:b], rl[None, :, a:b],
                        scale, causal)
        rdq[a:b], rdk[c:d], rdv[c:d] = g[0][0], g[1][0], g[2][0]
    runs = []
    for _ in range(2):
        out = torch.full((Tq, Hq, D), float("nan"), dtype=tq.dtype, device=dev)
        lse = torch.full((Hq, Tq), float("nan"), dtype=torch.float32, device=dev)
        # second launch: the one-item-per-workgroup shape used beside transfers (USP_LAUNCH_INTERLEAVE)
        _C.flash_fwd_packed(tq, tk, tv, sq, sk, max(lq), max(lk), scale, causal, lse, out=out, interleave=len(runs) == 1)
        runs.append((_f(out), _f(lse)))
    assert np.array_equal(runs[0][0], runs[1][0], equal_nan=True) and np.array_equal(runs[0][1], runs[1][1]), what
    fin = np.isfinite(rl)
    assert (np.isfinite(runs[0][1]) == fin).all(), what
    assert_close(runs[0][0], ro, *TOL[dt]["out"], what + " out")
    assert_close(runs[0][1][fin], rl[fin], 2e-3, 1e-4, what + " lse")
    lse_t = torch.from_numpy(np.ascontiguousarray(rl, dtype=np.float32)).to(dev)
    delta = torch.empty((Hq, Tq), dtype=torch.float32, device=dev)
    _C.bwd_delta(tdo[None], _t(o16, dt, dev)[None], delta[None])
    grads = []
    for _ in range(2):
        dq, dk, dv = (torch.full_like(t, float("nan")) for t in (tq, tk, tv))
        _C.flash_bwd_packed(tdo, tq, tk, tv, lse_t, delta, sq, sk, max(lq), max(lk), None, None, None, scale, causal,
                            dq16=dq, dk16=dk, dv16=dv, interleave=len(grads) == 1)
        grads.append([_f(x) for x in (dq, dk, dv)])
    for a_, b_, n_ in zip(grads[0], grads[1], ("dq", "dk", "dv")):
        assert np.array_equal(a_, b_), f"{what}: {n_} differs between two launches"
    for g_, r_, n_ in zip(grads[0], (rdq, rdk, rdv), ("dq", "dk", "dv")):
        assert_close(g_, r_, *TOL[dt]["grad"], f"{what} {n_}")
    assert int(_C.sched_block(dev).abs().sum()) == 0, "the work-queue control block must be left zeroed"
