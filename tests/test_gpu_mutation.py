"""Mutation tests: the GPU parity tests must be ABLE TO FAIL.

Every family of parity test (block kernels, packed mode, sliding window, K split / backward cuts, the virtual-rank
goldens, the sampled full-size checks) is run once more with the C-ABI binding patched so that, after the real
kernel has run, ONE element (or one whole row -- the image a NaN-prefilled buffer keeps when a kernel never writes
the row) of ONE result tensor is NaN.  The unmodified test function must then raise AssertionError.  A comparator
written as `bad = err > lim` passes all of these (NaN compares False); golden_util.assert_close does not.

The mutated position is (batch 0, LAST row, head 0, dim 0): every sampled test samples the last row / last key of
head 0, so the sampled families are covered by the same patch.
"""
import contextlib
import inspect

import numpy as np
import pytest
import torch

import test_gpu_parity as P
from golden_util import assert_close, close_mask

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import yunchang_amd  # noqa: F401
    from yunchang_amd import _C
    _C.load()
    return torch.device("cuda:0")


def _poison(t, kind):
    idx = (0,) * (t.dim() - 3) + (-1, 0)                  # (batch 0,) last row, head 0
    if kind == "elem":
        t[idx + (0,)] = float("nan")
    else:                                                 # "row": as if the kernel had never written the row
        t[idx] = float("nan")


@contextlib.contextmanager
def mutated(fn_name, arg_names, kind):
    """Patch yunchang_amd._C.<fn_name>: run the real launch, then poison the first result tensor among `arg_names`
    that the caller passed.  Yields a list that records the poisoned argument names (the test asserts it is
    non-empty: a mutation that never fired proves nothing)."""
    from yunchang_amd import _C
    real = getattr(_C, fn_name)
    sig = inspect.signature(real)
    fired = []

    def wrapper(*args, **kwargs):
        real(*args, **kwargs)
        bound = sig.bind(*args, **kwargs)
        for n in arg_names:
            t = bound.arguments.get(n)
            if t is not None:
                torch.cuda.current_stream().synchronize()
                _poison(t, kind)
                fired.append(n)
                break
    setattr(_C, fn_name, wrapper)
    try:
        yield fired
    finally:
        setattr(_C, fn_name, real)


FWD = ("flash_fwd", ("out", "acc"))
BWD_DQ = ("flash_bwd", ("dq16", "dq"))
BWD_DK = ("flash_bwd", ("dk16", "dk"))
BWD_DV = ("flash_bwd", ("dv16", "dv"))
PFWD = ("flash_fwd_packed", ("out", "acc"))
PBWD_DQ = ("flash_bwd_packed", ("dq16", "dq"))
PBWD_DV = ("flash_bwd_packed", ("dv16", "dv"))


def _must_fail(target, kind, fn, *args):
    with mutated(target[0], target[1], kind) as fired:
        with pytest.raises(AssertionError):
            fn(*args)
    assert fired, f"the mutation of {target} never fired in {fn.__name__}"


@pytest.mark.parametrize("kind", ["elem", "row"])
@pytest.mark.parametrize("target", [FWD, BWD_DQ, BWD_DK, BWD_DV], ids=["out", "dq", "dk", "dv"])
def test_block_family_fails_on_nan(dev, target, kind):
    _must_fail(target, kind, P.test_block_forward_backward_vs_oracle, dev, *P.SHAPES[1])
    _must_fail(target, kind, P.test_block_forward_backward_vs_oracle, dev, *P.SHAPES[3])      # ragged


@pytest.mark.parametrize("target,kind", [(FWD, "elem"), (FWD, "row"), (BWD_DQ, "elem"), (BWD_DK, "row"), (BWD_DV, "elem")],
                         ids=["out-elem", "out-row", "dq-elem", "dk-row", "dv-elem"])
def test_forced_row64_family_fails_on_nan(dev, target, kind):
    """The tests that pin the 64-row kernels (tests/test_gpu_row64.py) can fail too.  (Five of the eight (target, kind) pairs of
    round 5: each costs 5 s of oracle time in a suite with a time limit; both kinds stay covered on the forward and across the
    gradients.)"""
    import test_gpu_row64 as R
    _must_fail(target, kind, R.test_row64_edge_shapes, dev, *R.EDGE[4])          # ragged, bottom-right causal, GQA
    _must_fail(target, kind, R.test_row64_edge_shapes, dev, *R.EDGE[-1])         # several items per head
    if target is FWD:
        _must_fail(target, kind, R.test_row64_merge_in_and_partial_final_ranges, dev, *R.MERGE[1])


@pytest.mark.parametrize("target", [PFWD, PBWD_DQ, PBWD_DV], ids=["out", "dq", "dv"])
def test_packed_family_fails_on_nan(dev, target):
    _must_fail(target, "elem", P.test_packed_kernels_vs_oracle, dev, *P.PACKED[0])
    _must_fail(target, "row", P.test_packed_kernels_vs_oracle, dev, *P.PACKED[1])


@pytest.mark.parametrize("target", [FWD, BWD_DQ, BWD_DK], ids=["out", "dq", "dk"])
def test_window_family_fails_on_nan(dev, target):
    _must_fail(target, "elem", P.test_sliding_window_forward_backward_vs_oracle, dev, *P.WINDOWS[0])


def test_cut_families_fail_on_nan(dev):
    _must_fail(FWD, "elem", P.test_forward_k_split_through_the_binding, dev, 1, 1024, 1024, 2, 2, 128, True, "bfloat16")
    _must_fail(FWD, "row", P.test_forward_k_split_through_the_binding, dev, 1, 333, 200, 2, 1, 128, True, "bfloat16")
    for target in (BWD_DQ, BWD_DK, BWD_DV):
        _must_fail(target, "elem", P.test_backward_cuts_through_the_binding, dev, 1, 1024, 1024, 2, 2, 128, True,
                   "bfloat16")


@pytest.mark.parametrize("target", [FWD, BWD_DQ, BWD_DV], ids=["out", "dq", "dv"])
def test_virtual_rank_golden_family_fails_on_nan(dev, target):
    path = next(p for p in P.MULTI if P.Golden(p).bwd)
    _must_fail(target, "elem", P.test_multi_rank_golden_with_virtual_ranks, dev, path)


def test_varlen_golden_family_fails_on_nan(dev):
    from golden_util import varlen_golden_files
    _must_fail(PFWD, "elem", P.test_varlen_ring_golden_with_virtual_ranks, dev, varlen_golden_files()[0])


@pytest.mark.parametrize("target", [FWD, BWD_DQ, BWD_DK], ids=["out", "dq", "dk"])
def test_sampled_full_size_family_fails_on_nan(dev, target):
    """The NaN-prefilled buffers of this test were the reviewer's example: an unwritten row must not pass."""
    _must_fail(target, "row", P.test_c5_rank_block_shapes_against_sampled_fp64, dev, 16, 2)


def test_bench_sampled_parity_fails_on_nan(dev):
    """bench.sampled_parity (the 64K entry of the bench line): a NaN in a sampled row makes the figure NaN, and every
    gate written as `err < tol` is then False."""
    b = P._load_bench()
    t = b._fwd_bwd_kernels(1, 2048, 4, 2, 128, dev, 1, keep=True)["tensors"]
    clean = b.sampled_parity(t)["max_abs_err"]
    assert all(e == e for e in clean.values()) and clean["out"] < 2e-2 and clean["dq"] < 5e-2, clean
    for name in ("out", "dq", "dk", "dv"):
        u = dict(t)
        u[name] = t[name].clone()
        _poison(u[name], "elem")
        err = b.sampled_parity(u)["max_abs_err"]
        assert err[name] != err[name], (name, err)                       # NaN
        assert not (err[name] < 1.0)


def test_comparator_itself():
    """The comparator on host arrays (also covered without a GPU by tests/test_oracle_golden.py)."""
    want = np.array([1.0, -np.inf, 2.0])
    assert_close(np.array([1.0, -np.inf, 2.0]), want, 1e-3, 0, "same -inf passes")
    for got in ([np.nan, -np.inf, 2.0], [1.0, np.inf, 2.0], [1.0, np.nan, 2.0], [1.0, -np.inf, 3.0]):
        ok, _ = close_mask(np.array(got), want, 1e-3, 0)
        assert not ok.all()
        with pytest.raises(AssertionError):
            assert_close(np.array(got), want, 1e-3, 0)
