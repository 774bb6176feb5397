"""Helpers shared by the tests that read tests/golden/*.npz (made by tests/golden/make_golden.py
from the reference itself)."""
import glob
import os

import numpy as np

from oracle import usp_oracle as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Stated parity tolerances: max-abs envelopes of 16-bit USP attention vs an
# fp32/fp64 truth on N(0,1) inputs, with 2x head-room.  The reference's own bar is atol=1e-1 on the
# forward only (test/test_hybrid_attn.py:386).
TOL = {
    "bfloat16": dict(out=(2e-2, 2e-2), grad=(5e-2, 5e-2)),
    "float16": dict(out=(4e-3, 4e-3), grad=(1e-2, 1e-2)),
    "float32": dict(out=(2e-5, 2e-5), grad=(1e-4, 1e-4)),
}


def grad_tol(dtype: str, heads_summed: int = 1):
    """(atol, rtol) for a gradient: the stated tolerance, whatever the GQA group size.  (Rounds 1-3 widened atol by
    sqrt(G) for dK / dV sums over G query heads; the measured errors never needed it -- at G = 8, S = 65536 the sampled
    dK / dV errors are 2.1e-2 / 3.2e-2 against the un-widened 5e-2 -- so the gate is now what is measured.  The
    argument stays so that call sites keep saying which sums they check.)"""
    del heads_summed
    return TOL[dtype]["grad"]


def golden_files():
    """Dense fixtures (make_golden.py); the packed variable-length ones are varlen_golden_files()."""
    return sorted(f for f in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                  if not os.path.basename(f).startswith(("v_", "w_")))      # w_: sliding-window mask vectors


def varlen_golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "v_*.npz")))


def make_inputs_varlen(lens, Hq, Hkv, D, seed=0):
    """Must stay identical to tests/golden/make_golden_varlen.py:make_inputs_varlen."""
    T = int(sum(lens))
    rs = np.random.RandomState(seed)
    q = rs.standard_normal((T, Hq, D)).astype(np.float32)
    k = rs.standard_normal((T, Hkv, D)).astype(np.float32)
    v = rs.standard_normal((T, Hkv, D)).astype(np.float32)
    dout = rs.standard_normal((T, Hq, D)).astype(np.float32)
    return q, k, v, dout


def make_inputs(B, S, Hq, Hkv, D, seed=0):
    """Must stay identical to tests/golden/make_golden.py:make_inputs."""
    rs = np.random.RandomState(seed)
    q = rs.standard_normal((B, S, Hq, D)).astype(np.float32)
    k = rs.standard_normal((B, S, Hkv, D)).astype(np.float32)
    v = rs.standard_normal((B, S, Hkv, D)).astype(np.float32)
    dout = rs.standard_normal((B, S, Hq, D)).astype(np.float32)
    return q, k, v, dout


def round_to(x, dtype_s):
    """Round float32 values to the 16-bit dtype the fixture ran in (inputs were cast with
    torch .to(dtype) == round-to-nearest-even)."""
    if dtype_s == "bfloat16":
        return O.bf16_bits_to_f32(O.f32_to_bf16_bits(x))
    if dtype_s == "float16":
        return x.astype(np.float16).astype(np.float32)
    return x.astype(np.float32)


def decode(arr, dtype_s):
    if dtype_s == "bfloat16":
        return O.bf16_bits_to_f32(arr)
    if dtype_s == "float16":
        return arr.view(np.float16).astype(np.float32)
    return arr.astype(np.float32)


class Golden:
    def __init__(self, path):
        z = np.load(path)
        self.name = os.path.basename(path)[:-4]
        for key in ("ws", "ud", "rd", "B", "S", "Hq", "Hkv", "D", "seed"):
            setattr(self, key, int(z[key]))
        self.impl = str(z["impl"])
        self.layer = str(z["layer"]) if "layer" in z.files else "hybrid"   # hybrid | ulysses | qkvpacked
        self.dtype = str(z["dtype"])
        self.bwd = bool(z["bwd"])
        self.causal = bool(z["causal"])
        q, k, v, dout = make_inputs(self.B, self.S, self.Hq, self.Hkv, self.D, self.seed)
        self.q, self.k, self.v, self.dout = (round_to(t, self.dtype) for t in (q, k, v, dout))
        self.out = [decode(z[f"out_r{r}"], self.dtype) for r in range(self.ws)]
        if self.bwd:
            self.dq = [decode(z[f"dq_r{r}"], self.dtype) for r in range(self.ws)]
            self.dk = [decode(z[f"dk_r{r}"], self.dtype) for r in range(self.ws)]
            self.dv = [decode(z[f"dv_r{r}"], self.dtype) for r in range(self.ws)]

    def shard(self, x, rank):
        return O.EXTRACT[self.impl](x, rank, self.ws, self.rd, self.ud)


class VarlenGolden:
    """A v_*.npz fixture: the reference's (zigzag_)ring_flash_attn_varlen_func run on gloo."""

    def __init__(self, path):
        z = np.load(path)
        self.name = os.path.basename(path)[:-4]
        for key in ("ws", "Hq", "Hkv", "D", "seed"):
            setattr(self, key, int(z[key]))
        self.impl = str(z["impl"])                      # zigzag | basic
        self.dtype = str(z["dtype"])
        self.lens = [int(x) for x in z["lens"]]          # global sequence lengths
        self.cu = np.concatenate([[0], np.cumsum(self.lens)]).astype(np.int64)
        self.cu_local = self.cu // self.ws
        self.max_local = max(self.lens) // self.ws
        q, k, v, dout = make_inputs_varlen(self.lens, self.Hq, self.Hkv, self.D, self.seed)
        self.q, self.k, self.v, self.dout = (round_to(t, self.dtype) for t in (q, k, v, dout))
        for key in ("out", "dq", "dk", "dv"):
            setattr(self, key, [decode(z[f"{key}_r{r}"], self.dtype) for r in range(self.ws)])
        self.lse = [z[f"lse_r{r}"].astype(np.float32) for r in range(self.ws)]     # (H, T_local)

    def shard(self, x, rank):
        f = O.zigzag_extract_local_varlen if self.impl == "zigzag" else O.basic_extract_local_varlen
        return f(x, self.cu, rank, self.ws)

    def sim(self, douts=None):
        lq, lk, lv = ([self.shard(t, r) for r in range(self.ws)] for t in (self.q, self.k, self.v))
        f = O.zigzag_ring_varlen_sim if self.impl == "zigzag" else O.basic_ring_varlen_sim
        return f(lq, lk, lv, self.cu_local, douts)


def close_mask(got, want, atol, rtol):
    """Element-wise verdict of the parity check, written so that it CAN FAIL on NaN: an element passes only if
    `|got - want| <= atol + rtol*|want|` evaluates to True (False for NaN on either side), or if both sides are the SAME
    infinity (an LSE of -inf for a row without visible keys: `-inf - -inf` is NaN, but the values agree exactly).
    A NaN in `want` never passes -- the fixtures and the oracle hold none."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f"shape mismatch: got {got.shape}, want {want.shape}"
    same_inf = np.isinf(want) & (got == want)
    with np.errstate(invalid="ignore"):          # inf - inf: handled by same_inf, everything else must compare True
        err = np.abs(got - want)
        # an infinite `want` passes only for the identical infinity (rtol * inf = inf would otherwise admit anything)
        ok = np.where(np.isinf(want), same_inf, err <= atol + rtol * np.abs(want))
    return ok, np.where(same_inf, 0.0, err)


def assert_close(got, want, atol, rtol, what=""):
    """The one comparator of the parity tests (same semantics as the reference's only assertion,
    torch.testing.assert_close, test/test_hybrid_attn.py:386: NaN never equals anything)."""
    ok, err = close_mask(got, want, atol, rtol)
    if ok.all():
        return
    got = np.asarray(got, dtype=np.float64)
    n_nan = int(np.isnan(got).sum())
    finite = np.where(np.isnan(err), -1.0, err)
    raise AssertionError(f"{what}: {int((~ok).sum())} / {ok.size} elements out of tolerance (atol={atol}, rtol={rtol}); "
                         f"{n_nan} NaN in the result; max finite abs err {finite.max():.3e} "
                         f"at {np.unravel_index(finite.argmax(), finite.shape)}; first bad element at "
                         f"{np.unravel_index(int(np.argmax(~ok)), ok.shape)}")
