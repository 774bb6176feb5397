"""The one-wave-per-SIMD (64-row) kernel family under the driver-run parity suite.

`flash_fwd64_kernel` is the kernel the N = 1 bench line times, and by the library's own dispatch it only serves launches with
at least 256 work items: no golden fixture (S <= 1024) and no seed of the general fuzz sweep reaches it.  Here every call pins
the family PER CALL (ABI v6: `USP_FORCE_ROW64` in `usp_fwd_args.flags` / `usp_bwd_args.flags`, `family="row64"` in the
binding) and ASSERTS which kernels ran (`usp_last_launch_kinds()`), so that the ragged / Sq != Sk / empty-row / partial-final /
merge-in paths of that kernel (usp_flash_fwd64.hip: key ranges, MODE 1 / MODE 2 iterations, the wide and the narrow epilogue)
are checked against the CPU oracle on small shapes -- the reference's one assertion (test/test_hybrid_attn.py:386) against
`attention_ref` (test/test_utils.py:43-130), here with the stated tolerances of golden_util.TOL.

Every operand lives inside a NaN-filled arena (`_Arena`): the kernels address ragged tiles through raw-buffer descriptors whose
range check makes rows past the end read as zero (usp_mfma64.hpp: "HARDWARE ASSUMPTION"); a lane that escaped the clamp would
multiply a NaN into the result instead of whatever the allocator left behind the tensor.
"""
import os

import numpy as np
import pytest
import torch

from golden_util import TOL, assert_close, round_to
from oracle import usp_oracle as O

pytestmark = pytest.mark.gpu

_N_FWD = int(os.environ.get("USP_FUZZ_ROW64_FWD", "40"))        # larger sweeps: USP_FUZZ_ROW64_FWD=1000 (round 5 ran 100 by default: the GPU suite has a time limit)


@pytest.fixture(scope="module")
def dev():
    from yunchang_amd import _C
    _C.load()
    return torch.device("cuda:0")


def _f(t):
    return t.detach().float().cpu().numpy()


class _Arena:
    """16-bit operands carved out of ONE NaN-filled device buffer: 256 KiB of NaN in front of and 1 MiB behind every
    tensor (a 64-row tile that overshoots a tensor's end by its full height stays inside the poison at every stride used
    here).  The guard bands are checked after the launches: still NaN everywhere = nothing wrote outside its tensor."""
    FRONT, BACK = 128 * 1024, 512 * 1024          # elements (2 bytes each)

    def __init__(self, dt, dev, shapes):
        self.dt, self.dev = getattr(torch, dt), dev
        total = sum(self.FRONT + int(np.prod(s)) + self.BACK + 128 for s in shapes)
        self.buf = torch.full((total,), float("nan"), dtype=self.dt, device=dev)
        self.cur = 0
        self.used = []

    def put(self, x):
        n = x.size
        start = (self.cur + self.FRONT + 127) // 128 * 128          # 256-byte aligned
        view = self.buf[start:start + n].view(x.shape)
        view.copy_(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(self.dt))
        self.used.append((start, start + n))
        self.cur = start + n + self.BACK
        return view

    def out(self, shape):
        """A result tensor inside the arena, prefilled with NaN (an unwritten element stays NaN)."""
        return self.put(np.full(shape, np.nan, dtype=np.float32))

    def guards_intact(self):
        mask = torch.ones(self.buf.numel(), dtype=torch.bool, device=self.dev)
        for a, b in self.used:
            mask[a:b] = False
        return bool(torch.isnan(self.buf[mask]).all())


def _inputs(rs, B, Sq, Sk, Hq, Hkv, D, dt):
    return [round_to(rs.standard_normal(s).astype(np.float32), dt)
            for s in [(B, Sq, Hq, D), (B, Sk, Hkv, D), (B, Sk, Hkv, D), (B, Sq, Hq, D)]]


def _forward_forced(dev, case, rs, check_wave32=True):
    """Forward of one shape through the forced 64-row kernel, persistent and interleavable launch, against the oracle
    (and, for the record of what a family switch changes, against the 32-rows-per-wave family within tolerance)."""
    from yunchang_amd import _C
    B, Sq, Sk, Hq, Hkv, causal, dt = case
    D = 128
    what = f"row64 fwd B{B} Sq{Sq} Sk{Sk} Hq{Hq} Hkv{Hkv} causal={causal} {dt}"
    q, k, v, do = _inputs(rs, B, Sq, Sk, Hq, Hkv, D, dt)
    ar = _Arena(dt, dev, [q.shape, k.shape, v.shape, q.shape, q.shape, q.shape])
    tq, tk, tv = ar.put(q), ar.put(k), ar.put(v)
    scale = D ** -0.5
    ro, rl = O.attention_ref(q, k, v, causal, scale)
    runs = []
    for i in range(2):
        out = ar.out((B, Sq, Hq, D))
        lse = torch.full((B, Hq, Sq), float("nan"), dtype=torch.float32, device=dev)
        _C.flash_fwd(tq, tk, tv, scale, causal, lse, out=out, interleave=i == 1, family="row64", k_splits=0)
        assert _C.last_launch_kinds() == ("fwd_row64",), (what, _C.last_launch_kinds())
        runs.append((_f(out), _f(lse)))
    assert np.array_equal(runs[0][0], runs[1][0], equal_nan=True) and np.array_equal(runs[0][1], runs[1][1], equal_nan=True), \
        what + ": persistent and interleavable launches differ"
    got_o, got_l = runs[0]
    fin = np.isfinite(rl)
    assert (np.isfinite(got_l) == fin).all(), what + ": rows without a visible key must give lse = -inf"
    assert_close(got_o, ro, *TOL[dt]["out"], what + " out")
    assert_close(got_l[fin], rl[fin], 2e-3, 1e-4, what + " lse")
    assert (np.abs(got_o)[~np.broadcast_to(fin.transpose(0, 2, 1)[..., None], ro.shape)] == 0).all(), what + ": empty rows must be 0"
    assert ar.guards_intact(), what + ": a launch wrote outside its tensors"
    # the same shape with every query tile's keys cut into n work items (the split instantiation + the merge launch)
    n = int(rs.choice([2, 3, 4, 8]))
    out = ar.out((B, Sq, Hq, D))
    lse = torch.full((B, Hq, Sq), float("nan"), dtype=torch.float32, device=dev)
    _C.flash_fwd(tq, tk, tv, scale, causal, lse, out=out, interleave=bool(rs.randint(2)), family="row64", k_splits=n)
    assert _C.last_launch_kinds() == ("fwd_row64", "fwd_split_merge"), (what, n, _C.last_launch_kinds())
    cut_o, cut_l = _f(out), _f(lse)
    assert (np.isfinite(cut_l) == fin).all(), what + f" k_splits={n}: rows without a visible key must give lse = -inf"
    assert_close(cut_o, ro, *TOL[dt]["out"], what + f" out, k_splits={n}")
    assert_close(cut_l[fin], rl[fin], 2e-3, 1e-4, what + f" lse, k_splits={n}")
    assert (np.abs(cut_o)[~np.broadcast_to(fin.transpose(0, 2, 1)[..., None], ro.shape)] == 0).all(), what + f" k_splits={n}: empty rows must be 0"
    assert ar.guards_intact(), what + f" k_splits={n}: a launch wrote outside its tensors"
    if check_wave32:
        out8 = torch.empty((B, Sq, Hq, D), dtype=tq.dtype, device=dev)
        lse8 = torch.empty((B, Hq, Sq), dtype=torch.float32, device=dev)
        _C.flash_fwd(tq, tk, tv, scale, causal, lse8, out=out8, family="wave32", k_splits=0)
        kinds = _C.last_launch_kinds()
        assert kinds in (("fwd_wave8",), ("fwd_wave4",)), (what, kinds)
        assert_close(_f(out8), ro, *TOL[dt]["out"], what + " out (wave32 family)")
        assert_close(_f(lse8)[fin], got_l[fin], 1e-4, 1e-4, what + " lse: the two families")
    return (q, k, v, do, ro, rl, tq, tk, tv, ar)


def _backward_forced(dev, case, st):
    from yunchang_amd import _C
    B, Sq, Sk, Hq, Hkv, causal, dt = case
    D = 128
    what = f"row64 bwd B{B} Sq{Sq} Sk{Sk} Hq{Hq} Hkv{Hkv} causal={causal} {dt}"
    q, k, v, do, ro, rl, tq, tk, tv, ar0 = st
    scale = D ** -0.5
    o16 = round_to(ro.astype(np.float32), dt)
    rdq, rdk, rdv = O.block_bwd(do, q, k, v, o16, rl, scale, causal)
    ar = _Arena(dt, dev, [q.shape, q.shape, q.shape, k.shape, k.shape])
    tdo, to16 = ar.put(do), ar.put(o16)
    lse_t = torch.from_numpy(np.ascontiguousarray(rl, dtype=np.float32)).to(dev)
    delta = torch.empty((B, Hq, Sq), dtype=torch.float32, device=dev)
    _C.bwd_delta(tdo, to16, delta)
    dq, dk, dv = ar.out(q.shape), ar.out(k.shape), ar.out(k.shape)
    _C.flash_bwd(tdo, tq, tk, tv, lse_t, delta, None, None, None, scale, causal, dq16=dq, dk16=dk, dv16=dv,
                 family="row64", splits=(0, 0))
    kinds = set(_C.last_launch_kinds())
    assert {"dkdv_row64", "dq_row64"} <= kinds and not ({"dkdv_wave8", "dq_wave8"} & kinds), (what, kinds)
    for g_, r_, n_ in zip((dq, dk, dv), (rdq, rdk, rdv), ("dq", "dk", "dv")):
        atol, rtol = TOL[dt]["grad"]
        if (Sk if n_ == "dq" else Sq * (Hq // Hkv)) >= 1000:   # long sums of 16-bit-rounded products: test_gpu_fuzz._run_dense
            atol = max(atol, 8e-3 * float(np.sqrt(np.mean(np.square(r_, dtype=np.float64)))))
        assert_close(_f(g_), r_, atol, rtol, f"{what} {n_}")
    assert ar.guards_intact() and ar0.guards_intact(), what + ": a launch wrote outside its tensors"


# ------------------------------------------------------------------------------------------------
# edge shapes (the list of tests/test_gpu_parity.py::SHAPES at D = 128, plus tile-boundary cases)
# ------------------------------------------------------------------------------------------------
EDGE = [
    # B, Sq, Sk, Hq, Hkv, causal, dtype
    (1, 1, 1, 1, 1, True, "float16"),               # one score
    (1, 1, 700, 2, 1, False, "bfloat16"),           # one query row
    (2, 512, 512, 4, 4, True, "bfloat16"),
    (1, 384, 640, 4, 2, False, "bfloat16"),         # Sq != Sk, GQA
    (1, 200, 333, 3, 1, True, "bfloat16"),          # ragged, bottom-right causal
    (1, 333, 200, 2, 2, True, "float16"),           # rows with no visible key (the first 133)
    (1, 600, 70, 2, 1, True, "bfloat16"),           # whole 64-row waves and a whole 256-row item without any key
    (2, 77, 77, 2, 2, True, "bfloat16"),
    (1, 130, 1, 2, 1, False, "bfloat16"),           # one key
    (1, 64, 64, 1, 1, True, "bfloat16"),            # exactly one tile
    (1, 65, 63, 1, 1, True, "float16"),
    (1, 256, 256, 2, 2, True, "bfloat16"),          # exactly one item
    (1, 257, 255, 2, 1, True, "bfloat16"),          # one row into the second item
    (1, 255, 257, 2, 1, False, "float16"),
    (1, 1000, 40, 8, 1, False, "bfloat16"),         # fewer keys than a tile
    (3, 320, 320, 8, 8, True, "bfloat16"),          # 48 items: items % 8 == 0, the XCD-run walk
    (1, 1088, 1088, 8, 2, True, "bfloat16"),        # several items per head, diagonal in every wave position
]


@pytest.mark.parametrize("B,Sq,Sk,Hq,Hkv,causal,dt", EDGE)
def test_row64_edge_shapes(dev, B, Sq, Sk, Hq, Hkv, causal, dt):
    case = (B, Sq, Sk, Hq, Hkv, causal, dt)
    st = _forward_forced(dev, case, np.random.RandomState(7))
    _backward_forced(dev, case, st)


# ------------------------------------------------------------------------------------------------
# seeded sweep: ragged shapes forced through the 4 x 64 forward (every 4th seed: the backward too)
# ------------------------------------------------------------------------------------------------
def _fuzz_case(rs):
    dt = str(rs.choice(["bfloat16", "bfloat16", "float16"]))
    Hkv = int(rs.choice([1, 2]))
    Hq = Hkv * int(rs.choice([1, 2, 4]))
    B = int(rs.choice([1, 2]))
    draw = lambda: int(rs.choice([rs.randint(1, 130), rs.randint(130, 520), rs.randint(520, 900)]))
    Sq = draw()
    Sk = Sq if rs.rand() < 0.5 else draw()
    causal = bool(rs.rand() < 0.6)
    return B, Sq, Sk, Hq, Hkv, causal, dt


@pytest.mark.parametrize("seed", range(_N_FWD))
def test_fuzz_row64_forward_forced(dev, seed):
    rs = np.random.RandomState(7000 + seed)
    case = _fuzz_case(rs)
    st = _forward_forced(dev, case, rs, check_wave32=seed % 2 == 0)
    if seed % 4 == 0:
        _backward_forced(dev, case, st)


# ------------------------------------------------------------------------------------------------
# the fused ring merge of the 64-row forward: merge-in, partial final ranges, untouched rows
# ------------------------------------------------------------------------------------------------
MERGE = [
    # Sq, Sk, keys of the first call, causal second call, final_begin, final_end, dtype
    (300, 812, 512, True, 100, 257, "bfloat16"),    # q x k[:512] full, then q x k[512:] causal (bottom-right): = causal over all
    (300, 812, 512, True, 0, 300, "float16"),       # everything final in the second call: the narrow epilogue + merge
    (520, 700, 333, False, 64, 512, "bfloat16"),    # both calls full attention, ragged split
    (256, 128, 64, False, 0, 0, "bfloat16"),        # nothing final: fp32 accumulator only
    (130, 390, 260, True, 129, 130, "bfloat16"),    # one final row
]


@pytest.mark.parametrize("Sq,Sk,Sa,causal2,fb,fe,dt", MERGE)
def test_row64_merge_in_and_partial_final_ranges(dev, Sq, Sk, Sa, causal2, fb, fe, dt):
    """Two calls over a split of the keys == one call over all of them (update_out_and_lse, yunchang/ring/utils.py:10-51,
    fused into the epilogue): the first leaves (acc, lse) in fp32, the second merges, emits rows [fb, fe) in 16 bits and the
    others as fp32 -- and touches nothing else."""
    from yunchang_amd import _C
    B, Hq, Hkv, D = 2, 4, 2, 128
    rs = np.random.RandomState(11)
    q, k, v, _ = _inputs(rs, B, Sq, Sk, Hq, Hkv, D, dt)
    if causal2:
        assert Sk - Sa == Sq and Sa <= Sk - Sq + 1
    ro, rl = O.attention_ref(q, k, v, causal2, D ** -0.5)
    ar = _Arena(dt, dev, [q.shape, k.shape, v.shape, q.shape])
    tq, tk, tv = ar.put(q), ar.put(k), ar.put(v)
    out = ar.out(q.shape)
    acc = torch.full((B, Sq, Hq, D), float("nan"), dtype=torch.float32, device=dev)
    lse = torch.full((B, Hq, Sq), float("nan"), dtype=torch.float32, device=dev)
    scale = D ** -0.5
    _C.flash_fwd(tq, tk[:, :Sa], tv[:, :Sa], scale, False, lse, out=None, acc=acc, final_begin=0, final_end=0, family="row64")
    assert _C.last_launch_kinds() == ("fwd_row64",)
    assert torch.isfinite(acc).all() and torch.isfinite(lse).all()
    acc1 = acc.clone()
    _C.flash_fwd(tq, tk[:, Sa:], tv[:, Sa:], scale, causal2, lse, out=out, acc=acc, merge_in=True, final_begin=fb, final_end=fe,
                 family="row64")
    assert _C.last_launch_kinds() == ("fwd_row64",)
    what = f"merge Sq{Sq} Sk{Sk}={Sa}+{Sk - Sa} final [{fb},{fe}) {dt}"
    assert_close(_f(lse), rl, 2e-3, 1e-4, what + " lse")
    o, a = _f(out), _f(acc)
    fin = np.zeros(Sq, dtype=bool)
    fin[fb:fe] = True
    assert_close(o[:, fin], ro[:, fin], *TOL[dt]["out"], what + " final rows")
    assert np.isnan(o[:, ~fin]).all(), what + ": rows outside the final range must not be written to `out`"
    assert_close(a[:, ~fin], ro[:, ~fin], 2e-3, 2e-3, what + " running rows (fp32)")
    assert np.array_equal(a[:, fin], _f(acc1)[:, fin]), what + ": accumulator rows of the final range must not be rewritten"
    assert ar.guards_intact()
    # the same second call with its keys cut into 3 work items: the cuts go to the workspace, split_merge_kernel merges them
    # with the running result and emits the same final / running rows
    out2 = torch.full_like(out, float("nan"))
    acc2, lse2 = acc1.clone(), torch.full_like(lse, float("nan"))
    _C.flash_fwd(tq, tk[:, :Sa], tv[:, :Sa], scale, False, lse2, out=None, acc=acc2, final_begin=0, final_end=0, family="row64",
                 k_splits=0)
    _C.flash_fwd(tq, tk[:, Sa:], tv[:, Sa:], scale, causal2, lse2, out=out2, acc=acc2, merge_in=True, final_begin=fb, final_end=fe,
                 family="row64", k_splits=3)
    assert _C.last_launch_kinds() == ("fwd_row64", "fwd_split_merge")
    assert_close(_f(lse2), rl, 2e-3, 1e-4, what + " lse (3 cuts)")
    o2, a2 = _f(out2), _f(acc2)
    assert_close(o2[:, fin], ro[:, fin], *TOL[dt]["out"], what + " final rows (3 cuts)")
    assert np.isnan(o2[:, ~fin]).all(), what + " (3 cuts): rows outside the final range must not be written to `out`"
    assert_close(a2[:, ~fin], ro[:, ~fin], 2e-3, 2e-3, what + " running rows (fp32, 3 cuts)")
    assert np.array_equal(a2[:, fin], _f(acc1)[:, fin]), what + " (3 cuts): accumulator rows of the final range must not be rewritten"
    assert ar.guards_intact()


@pytest.mark.parametrize("B,Sq,Sk,Hq,Hkv,causal,cuts", [(1, 1024, 1024, 2, 2, True, (2, 2)), (1, 333, 200, 2, 1, True, (3, 4)),
                                                        (2, 300, 712, 4, 2, False, (8, 2)), (1, 2048, 2048, 2, 1, True, (4, 1)),
                                                        (1, 1500, 1500, 1, 1, True, (5, 0))])
def test_row64_backward_cuts(dev, B, Sq, Sk, Hq, Hkv, causal, cuts):
    """The cuts of few-item backward launches (ABI v5 dq_splits / dkdv_splits) through the 64-row kernels: the dQ kernel
    takes the key cut since round 5 (`flash_bwd_dq64_kernel` + `reduce_cuts_kernel`; round 4 fell back to the 8-wave dQ
    kernel), the dK/dV kernel the query cut.  Every cut against the uncut forced launch (same sums in another order: tight)
    and against the oracle; fp32-accumulated dQ on top of a previous value as a ring step does it."""
    from yunchang_amd import _C
    dt, D = "bfloat16", 128
    rs = np.random.RandomState(31)
    q, k, v, do = _inputs(rs, B, Sq, Sk, Hq, Hkv, D, dt)
    scale = D ** -0.5
    ro, rl = O.attention_ref(q, k, v, causal, scale)
    o16 = round_to(ro.astype(np.float32), dt)
    rdq, rdk, rdv = O.block_bwd(do, q, k, v, o16, rl, scale, causal)
    tq, tk, tv, tdo, to16 = (torch.from_numpy(x).to(torch.bfloat16).to(dev) for x in (q, k, v, do, o16))
    lse_t = torch.from_numpy(np.ascontiguousarray(rl, dtype=np.float32)).to(dev)
    delta = torch.empty((B, Hq, Sq), dtype=torch.float32, device=dev)
    _C.bwd_delta(tdo, to16, delta)
    res = {}
    for name, sp in (("uncut", (0, 0)), ("cut", cuts)):
        dq, dk, dv = (torch.full_like(t, float("nan")) for t in (tq, tk, tv))
        _C.flash_bwd(tdo, tq, tk, tv, lse_t, delta, None, None, None, scale, causal, dq16=dq, dk16=dk, dv16=dv, family="row64", splits=sp)
        kinds = set(_C.last_launch_kinds())
        assert {"dkdv_row64", "dq_row64"} <= kinds and not ({"dkdv_wave8", "dq_wave8"} & kinds), (name, kinds)
        if name == "cut":
            assert ("reduce_cuts" in kinds) == (cuts[0] > 1) and ("reduce_heads" in kinds) == (cuts[1] > 1 or Hq > Hkv), kinds
        res[name] = [_f(x) for x in (dq, dk, dv)]
    for a_, b_, r_, n_ in zip(res["cut"], res["uncut"], (rdq, rdk, rdv), ("dq", "dk", "dv")):
        assert_close(a_, b_, 2e-2, 2e-2, f"{n_}: cut {cuts} against the uncut launch")          # (one 16-bit rounding apart)
        assert_close(a_, r_, *TOL[dt]["grad"], f"{n_}: cut {cuts} against the oracle")
    # fp32 accumulation on top of a previous value (accum_dq: what a ring step asks for), cut along the keys
    if cuts[0] > 1:
        base = torch.randn(B, Sq, Hq, D, device=dev, dtype=torch.float32)
        acc = base.clone()
        dk32, dv32 = (torch.empty(B, Sk, Hkv, D, device=dev, dtype=torch.float32) for _ in range(2))
        _C.flash_bwd(tdo, tq, tk, tv, lse_t, delta, acc, dk32, dv32, scale, causal, accum_dq=True, family="row64", splits=cuts)
        assert "reduce_cuts" in _C.last_launch_kinds()
        assert_close(_f(acc) - _f(base), rdq, *TOL[dt]["grad"], "accumulated dq with a key cut")


def test_forced_family_refuses_what_it_does_not_serve(dev):
    """USP_FORCE_ROW64 on a call the family does not serve returns USP_EUNSUPPORTED (RuntimeError in the binding) and
    launches nothing -- a test that pins the family can never silently run on the other one."""
    from yunchang_amd import _C
    q = torch.randn(1, 300, 2, 64, device=dev).bfloat16()
    lse = torch.empty(1, 2, 300, dtype=torch.float32, device=dev)
    out = torch.full_like(q, float("nan"))
    with pytest.raises(RuntimeError, match="unsupported"):
        _C.flash_fwd(q, q, q, 0.125, True, lse, out=out, family="row64")          # head dim 64
    assert _C.last_launch_kinds() == () and torch.isnan(out).all()
    q = torch.randn(1, 300, 2, 128, device=dev).bfloat16()
    out = torch.full_like(q, float("nan"))
    with pytest.raises(RuntimeError, match="unsupported"):
        _C.flash_fwd(q, q, q, 0.09, True, lse, out=out, family="row64", window=(17, 0))
    assert torch.isnan(out).all()
    _C.flash_fwd(q, q, q, 0.09, True, lse, out=out, family="row64", k_splits=2)   # (round 5: the family serves K splits)
    assert _C.last_launch_kinds() == ("fwd_row64", "fwd_split_merge") and torch.isfinite(out).all()
    out.fill_(float("nan"))
    _C.flash_fwd(q, q, q, 0.09, True, lse, out=out, family="wave32")
    assert _C.last_launch_kinds() == ("fwd_wave4",) and torch.isfinite(out).all()
    # the in-process default (what bench.py's same-box A/B uses)
    assert _C.set_kernel_family("row64") == "auto"
    try:
        _C.flash_fwd(q, q, q, 0.09, True, lse, out=out)
        assert _C.last_launch_kinds() == ("fwd_row64",)
    finally:
        _C.set_kernel_family("auto")
    _C.flash_fwd(q, q, q, 0.09, True, lse, out=out)
    assert _C.last_launch_kinds() == ("fwd_wave4",)           # 4 items: the library's own choice for a small launch


def test_default_dispatch_at_the_bench_shapes(dev):
    """What the N = 1 bench line times: at C2 (B2 S8192 H16 D128 causal) the library itself picks the 64-row family for all
    three flash kernels."""
    from yunchang_amd import _C
    B, S, H, D = 2, 8192, 16, 128
    q, k, v, do = (torch.randn(B, S, H, D, device=dev).bfloat16() for _ in range(4))
    out = torch.empty_like(q)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    _C.flash_fwd(q, k, v, D ** -0.5, True, lse, out=out)
    assert _C.last_launch_kinds() == ("fwd_row64",)
    delta = torch.empty_like(lse)
    _C.bwd_delta(do, out, delta)
    dq, dk, dv = (torch.empty_like(t) for t in (q, k, v))
    _C.flash_bwd(do, q, k, v, lse, delta, None, None, None, D ** -0.5, True, dq16=dq, dk16=dk, dv16=dv)
    assert _C.last_launch_kinds() == ("dkdv_row64", "dq_row64")


# ------------------------------------------------------------------------------------------------
# the gradient tolerance floor of the fuzz sweeps, pinned (review of round 4, "a tolerance was widened")
# ------------------------------------------------------------------------------------------------
def _bwd_16bit_model(do, q, k, v, o16, lse, scale, causal, dt, prescale_k):
    """fp64 restatement of the block backward WITH the two roundings every 16-bit flash backward performs: P is rounded
    to the 16-bit type before dV = P^T dO (and dS is formed from that rounded P), dS is rounded before dQ = dS K and
    dK = dS^T Q.  `prescale_k`: the 64-row dK/dV kernel also rounds K * scale * log2(e) once per item (usp_flash_bwd64.hip)."""
    B, Sq, Hq, D = q.shape
    Sk, Hkv = k.shape[1], k.shape[2]
    g = Hq // Hkv
    kk, vv = np.repeat(k, g, axis=2).astype(np.float64), np.repeat(v, g, axis=2).astype(np.float64)
    qd, dod = q.astype(np.float64), do.astype(np.float64)
    log2e = 1.4426950408889634
    if prescale_k:
        k2 = round_to((kk * (scale * log2e)).astype(np.float32), dt).astype(np.float64)
        s2 = np.einsum("bthd,bshd->bhts", qd, k2, optimize=True)                 # exponent, base 2
    else:
        s2 = np.einsum("bthd,bshd->bhts", qd, kk, optimize=True) * (scale * log2e)
    if causal:
        row, col = np.arange(Sq)[:, None], np.arange(Sk)[None, :]
        s2 = np.where(col > row + Sk - Sq, -np.inf, s2)
    lse_safe = np.where(np.isfinite(lse), lse, 0.0)
    p = np.where(np.isfinite(lse)[..., None], np.exp2(s2 - lse_safe[..., None] * log2e), 0.0)
    p16 = round_to(p.astype(np.float32), dt).astype(np.float64)
    dv = np.einsum("bhts,bthd->bshd", p16, dod, optimize=True).reshape(B, Sk, Hkv, g, D).sum(3)
    dp = np.einsum("bthd,bshd->bhts", dod, vv, optimize=True)
    delta = np.einsum("bthd,bthd->bht", dod, o16.astype(np.float64))
    ds16 = round_to((p16 * (dp - delta[..., None])).astype(np.float32), dt).astype(np.float64)
    dq = np.einsum("bhts,bshd->bthd", ds16, kk, optimize=True) * scale
    dk = (np.einsum("bhts,bthd->bshd", ds16, qd, optimize=True) * scale).reshape(B, Sk, Hkv, g, D).sum(3)
    return dq, dk, dv


def test_long_sum_gradient_noise_is_that_of_16bit_products(dev):
    """Seed 5000+x of the round-4 sweep (B2 Sq1191 Sk10 Hq8 Hkv2 D128, full attention) misses `atol + rtol |want|` on dK:
    every dK entry sums Sq * G = 4764 products of 16-bit-rounded factors (ten keys: P is not small), so its absolute
    error is ~1e-3 x rms(entry) per sigma whatever its own value, and entries near a zero crossing of a tensor whose
    entries are ~20 fail a bound that scales with |want|.  Pinned here, instead of argued in a comment:
      (1) BOTH kernel families miss the un-floored bound on this case, and by the same amount (within 35 %);
      (2) against the same arithmetic with P and dS rounded to 16 bits (`_bwd_16bit_model`) both pass the UN-FLOORED
          bound with half of it to spare -- the miss is the rounding of the products, not the kernels' sums;
      (3) the 64-row dK/dV kernel's extra rounding (K * scale * log2 e in bf16) stays below 1.6x the other family's dV
          error against exact arithmetic (round 4 measured 3.2e-2 -> 5.3e-2 at S = 65536, G = 8; this is its guard)."""
    from yunchang_amd import _C
    from golden_util import close_mask
    B, Sq, Sk, Hq, Hkv, D, causal, dt = 2, 1191, 10, 8, 2, 128, False, "bfloat16"
    rs = np.random.RandomState(5)
    q, k, v, do = _inputs(rs, B, Sq, Sk, Hq, Hkv, D, dt)
    scale = D ** -0.5
    ro, rl = O.attention_ref(q, k, v, causal, scale)
    o16 = round_to(ro.astype(np.float32), dt)
    exact = O.block_bwd(do, q, k, v, o16, rl, scale, causal)
    tq, tk, tv, tdo, to16 = (torch.from_numpy(x).to(torch.bfloat16).to(dev) for x in (q, k, v, do, o16))
    lse_t = torch.from_numpy(np.ascontiguousarray(rl, dtype=np.float32)).to(dev)
    delta = torch.empty((B, Hq, Sq), dtype=torch.float32, device=dev)
    _C.bwd_delta(tdo, to16, delta)
    atol, rtol = TOL[dt]["grad"]
    err, got = {}, {}
    for fam in ("row64", "wave32"):
        dq, dk, dv = (torch.full_like(t, float("nan")) for t in (tq, tk, tv))
        _C.flash_bwd(tdo, tq, tk, tv, lse_t, delta, None, None, None, scale, causal, dq16=dq, dk16=dk, dv16=dv, family=fam,
                     splits=(0, 0))
        kinds = set(_C.last_launch_kinds())
        assert ({"dkdv_row64", "dq_row64"} <= kinds) if fam == "row64" else ({"dkdv_wave8", "dq_wave8"} <= kinds), (fam, kinds)
        got[fam] = [_f(x) for x in (dq, dk, dv)]
        err[fam] = [float(np.abs(g_ - e_).max()) for g_, e_ in zip(got[fam], exact)]
    # (1) the un-floored bound fails for dK in both families, by the same amount
    miss = {}
    for fam in got:
        ok, e = close_mask(got[fam][1], exact[1], atol, rtol)
        miss[fam] = float((e - (atol + rtol * np.abs(exact[1]))).max())
    assert miss["row64"] > 0 and miss["wave32"] > 0, f"the case no longer misses the un-floored bound: {miss} -- drop the floor"
    assert abs(miss["row64"] - miss["wave32"]) <= 0.35 * max(miss.values()), miss
    # (2) against 16-bit-rounded products both pass the un-floored bound, with room
    for fam in got:
        model = _bwd_16bit_model(do, q, k, v, o16, rl, scale, causal, dt, prescale_k=fam == "row64")
        for g_, m_, n_ in zip(got[fam], model, ("dq", "dk", "dv")):
            assert_close(g_, m_, atol / 2, rtol / 2, f"{fam} {n_} against the 16-bit-product model")
    # (3) the pre-scaled K of the 64-row dK/dV kernel
    assert err["row64"][2] <= 1.6 * err["wave32"][2] + 1e-3, err
