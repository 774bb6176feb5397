"""Pins the CPU oracle (oracle/usp_oracle.py) against outputs of the reference itself
(tests/golden/*.npz, produced by tests/golden/make_golden.py on CPU/gloo)."""
import os

import numpy as np
import pytest

from oracle import usp_oracle as O
from golden_util import Golden, TOL, VarlenGolden, assert_close, golden_files, varlen_golden_files

FILES = golden_files()


def test_fixtures_present():
    assert len(FILES) >= 8, "golden fixtures missing: run tests/golden/make_golden.py in the build container"


@pytest.mark.parametrize("path", FILES, ids=lambda p: p.split("/")[-1][:-4])
def test_oracle_matches_reference_run(path):
    g = Golden(path)
    lq = [g.shard(g.q, r) for r in range(g.ws)]
    lk = [g.shard(g.k, r) for r in range(g.ws)]
    lv = [g.shard(g.v, r) for r in range(g.ws)]
    outs, ctx = O.usp_forward_sim(lq, lk, lv, g.ud, g.rd, g.impl, causal=g.causal, return_ctx=True)
    atol, rtol = TOL[g.dtype]["out"]
    for r in range(g.ws):
        assert outs[r].shape == g.out[r].shape
        assert_close(g.out[r], outs[r], atol, rtol, f"{g.name} out rank {r}")
    if g.bwd:
        ldo = [g.shard(g.dout, r) for r in range(g.ws)]
        dqs, dks, dvs = O.usp_backward_sim(ldo, ctx, g.ud, g.rd, g.impl, causal=g.causal)
        atol, rtol = TOL[g.dtype]["grad"]
        for r in range(g.ws):
            assert_close(g.dq[r], dqs[r], atol, rtol, f"{g.name} dq rank {r}")
            assert_close(g.dk[r], dks[r], atol, rtol, f"{g.name} dk rank {r}")
            assert_close(g.dv[r], dvs[r], atol, rtol, f"{g.name} dv rank {r}")


@pytest.mark.parametrize("path", [f for f in FILES if "c1_" in f or "c5_w8_u2r4_gqa_bf16" in f],
                         ids=lambda p: p.split("/")[-1][:-4])
def test_sharded_sim_equals_global_attention(path):
    """Size-independent property: USP over any grid == plain attention on the unsharded tensors."""
    g = Golden(path)
    full, _ = O.attention_ref(g.q, g.k, g.v, causal=True)
    lq = [g.shard(g.q, r) for r in range(g.ws)]
    lk = [g.shard(g.k, r) for r in range(g.ws)]
    lv = [g.shard(g.v, r) for r in range(g.ws)]
    outs = O.usp_forward_sim(lq, lk, lv, g.ud, g.rd, g.impl, causal=True)
    for r in range(g.ws):
        np.testing.assert_allclose(outs[r], g.shard(full, r), atol=1e-12, rtol=1e-10)


VFILES = varlen_golden_files()


def test_varlen_fixtures_present():
    assert len(VFILES) >= 4, "run tests/golden/make_golden_varlen.py in the build container"


@pytest.mark.parametrize("path", VFILES, ids=lambda p: p.split("/")[-1][:-4])
def test_varlen_oracle_matches_reference_run(path):
    """Packed variable-length ring schedules: oracle vs the reference's own run (out, lse, dq, dk, dv)."""
    g = VarlenGolden(path)
    ldo = [g.shard(g.dout, r) for r in range(g.ws)]
    outs, lses, dqs, dks, dvs = g.sim(ldo)
    atol, rtol = TOL[g.dtype]["out"]
    gt, gr = TOL[g.dtype]["grad"]
    for r in range(g.ws):
        assert outs[r].shape == g.out[r].shape and lses[r].shape == g.lse[r].shape
        assert_close(g.out[r], outs[r], atol, rtol, f"{g.name} out rank {r}")
        assert_close(g.lse[r], lses[r], atol, rtol, f"{g.name} lse rank {r}")
        assert_close(g.dq[r], dqs[r], gt, gr, f"{g.name} dq rank {r}")
        assert_close(g.dk[r], dks[r], gt, gr, f"{g.name} dk rank {r}")
        assert_close(g.dv[r], dvs[r], gt, gr, f"{g.name} dv rank {r}")


@pytest.mark.parametrize("path", VFILES[:2], ids=lambda p: p.split("/")[-1][:-4])
def test_varlen_sharded_sim_equals_per_sequence_attention(path):
    """Size-independent property: the ring schedule over shards == attention on each whole sequence."""
    g = VarlenGolden(path)
    full, full_lse = O.varlen_attention_ref(g.q, g.k, g.v, g.cu, causal=True)
    outs, lses, *_ = g.sim()
    for r in range(g.ws):
        np.testing.assert_allclose(outs[r], g.shard(full, r), atol=1e-12, rtol=1e-10)
        np.testing.assert_allclose(lses[r], g.shard(full_lse.T, r).T, atol=1e-12, rtol=1e-10)


@pytest.mark.parametrize("path", [f for f in golden_files() if Golden(f).ws == 1], ids=lambda p: p.split("/")[-1][:-4])
def test_cpu_baseline_port_equals_the_reference_single_rank_run(path):
    """oracle/ref_cpu_path.py (bench.py's `cpu_baseline`, kind "port": the reference cannot be imported on the GPU box)
    against the reference's OWN single-rank runs (fixtures made by tests/golden/make_golden.py importing the reference
    with the same CPU attention op in the efficient op's place): the same calls on the same data must give the same
    bits -- C1 in fp32 and bf16, and the B = 2 fixture of configs[1]'s topology."""
    import torch
    from oracle import ref_cpu_path as R
    g = Golden(path)
    dtype = getattr(torch, g.dtype)
    q, k, v = (torch.from_numpy(x).to(dtype) for x in (g.q, g.k, g.v))
    out, _ = R.long_context_attention_forward_cpu(q, k, v, g.causal)
    got = out.float().numpy()
    assert got.shape == g.out[0].shape
    assert np.array_equal(got, g.out[0]), f"{g.name}: max abs diff {np.abs(got - g.out[0]).max():.3e}"


def test_oracle_window_mask_equals_the_reference():
    """oracle.attention_ref(window=...) against the reference's own attention_ref with `window_size`
    (tests/golden/w_window_ref.npz, made by tests/golden/make_golden_window.py importing the reference)."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "w_window_ref.npz"))
    for i in range(int(z["n"])):
        Sq, Sk, Hq, Hkv, D, causal, wl, wr = (int(x) for x in z[f"case{i}"])
        rs = np.random.RandomState(100 + i)
        q, k, v = (rs.standard_normal((1, S, H, D)).astype(np.float32) for S, H in ((Sq, Hq), (Sk, Hkv), (Sk, Hkv)))
        out, _ = O.attention_ref(q, k, v, causal=bool(causal), window=(wl, wr))
        assert_close(out, z[f"out{i}"], 2e-5, 2e-5, f"window case {i}: {(Sq, Sk, causal, wl, wr)}")


def test_the_comparator_can_fail():
    """golden_util.assert_close is the one comparator of every parity test: NaN on the result side fails, an
    unwritten (NaN-prefilled) row fails, the SAME infinity on both sides passes (an LSE of -inf), opposite
    infinities and inf-vs-finite fail, and no RuntimeWarning escapes (`-inf - -inf`)."""
    import warnings
    from golden_util import close_mask
    want = np.array([[1.0, -np.inf, 2.0], [0.0, 0.5, -3.0]])
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert_close(want.copy(), want, 1e-6, 0.0, "identical")
        for i, bad in (((0, 0), np.nan), ((0, 1), np.inf), ((0, 1), 0.0), ((0, 1), np.nan), ((1, 2), -3.1), ((1, 0), np.inf)):
            got = want.copy()
            got[i] = bad
            ok, _ = close_mask(got, want, 1e-3, 1e-3)
            assert not ok[i] and ok.sum() == ok.size - 1
            with pytest.raises(AssertionError):
                assert_close(got, want, 1e-3, 1e-3, "mutated")
        row = want.copy()
        row[1] = np.nan                                           # a row the kernel never wrote
        with pytest.raises(AssertionError, match="3 NaN"):
            assert_close(row, want, 1e-3, 1e-3, "unwritten row")
        with pytest.raises(AssertionError):
            assert_close(np.full_like(want, np.nan), want, 1e-3, 1e-3, "all NaN")
        with pytest.raises(AssertionError):
            assert_close(want[:1], want, 1e-3, 1e-3, "shape")
