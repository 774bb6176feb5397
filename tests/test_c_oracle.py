"""The C restatement (oracle/attn_oracle.c) must equal the numpy oracle (pinned to reference runs)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import usp_oracle as O

ODIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(ODIR, "libattn_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", ODIR])
    L = ctypes.CDLL(so)
    return L


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


@pytest.mark.parametrize("B,Sq,Sk,Hq,Hkv,D,causal", [
    (1, 1, 1, 1, 1, 32, 1), (2, 33, 33, 2, 2, 32, 1), (1, 40, 70, 4, 2, 64, 1), (1, 70, 40, 2, 1, 32, 1),
    (1, 64, 96, 2, 2, 128, 0),
])
def test_c_oracle_equals_numpy(lib, B, Sq, Sk, Hq, Hkv, D, causal):
    rs = np.random.RandomState(5)
    q = rs.standard_normal((B, Sq, Hq, D)).astype(np.float32)
    k = rs.standard_normal((B, Sk, Hkv, D)).astype(np.float32)
    v = rs.standard_normal((B, Sk, Hkv, D)).astype(np.float32)
    do = rs.standard_normal((B, Sq, Hq, D)).astype(np.float32)
    scale = D ** -0.5
    out = np.empty_like(q); lse = np.empty((B, Hq, Sq), np.float32)
    lib.usp_oracle_attn_fwd(_p(q), _p(k), _p(v), B, Sq, Sk, Hq, Hkv, D, ctypes.c_float(scale), causal, _p(out), _p(lse))
    ro, rl = O.attention_ref(q, k, v, causal=bool(causal), softmax_scale=scale)
    np.testing.assert_allclose(out, ro, atol=2e-6, rtol=1e-5)
    fin = np.isfinite(rl)
    assert (np.isfinite(lse) == fin).all()
    np.testing.assert_allclose(lse[fin], rl[fin], atol=2e-6, rtol=1e-5)
    dq = np.empty_like(q); dk = np.empty_like(k); dv = np.empty_like(v)
    lib.usp_oracle_attn_bwd(_p(do), _p(q), _p(k), _p(v), _p(out), _p(lse), B, Sq, Sk, Hq, Hkv, D,
                            ctypes.c_float(scale), causal, _p(dq), _p(dk), _p(dv))
    rdq, rdk, rdv = O.block_bwd(do, q, k, v, ro, rl, scale, bool(causal))
    np.testing.assert_allclose(dq, rdq, atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(dk, rdk, atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(dv, rdv, atol=2e-5, rtol=1e-4)
