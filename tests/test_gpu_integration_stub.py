"""The binding INTEGRATION.md shows a maintainer of the reference (section B: a ctypes stub next to
yunchang/kernels/attention.py) is executed VERBATIM from the document and checked against the CPU oracle: forward and
backward through the C ABI with nothing of this package's Python in between."""
import os
import re

import numpy as np
import pytest
import torch

from golden_util import TOL, assert_close, grad_tol, round_to
from oracle import usp_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_namespace():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n# yunchang/kernels/usp_hip.py.*?\n(.*?)```", md, flags=re.S).group(1)
    block = block.replace("/path/to/long-context-attention_amd/libusp_hip.so",
                          os.path.join(ROOT, "long-context-attention_amd", "libusp_hip.so"))
    ns = {}
    exec(compile(block, "INTEGRATION.md:usp_hip.py", "exec"), ns)
    return ns


@pytest.mark.parametrize("B,S,Hq,Hkv,D,causal,dt", [(2, 320, 4, 4, 128, True, "bfloat16"), (1, 257, 8, 2, 64, True, "float16"),
                                                    (1, 192, 2, 2, 128, False, "bfloat16")])
def test_documented_stub_forward_and_backward(B, S, Hq, Hkv, D, causal, dt):
    ns = _stub_namespace()
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(7)
    q, k, v, do = (round_to(rs.standard_normal(s).astype(np.float32), dt) for s in
                   [(B, S, Hq, D), (B, S, Hkv, D), (B, S, Hkv, D), (B, S, Hq, D)])
    tq, tk, tv, tdo = (torch.from_numpy(x).to(getattr(torch, dt)).to(dev) for x in (q, k, v, do))
    out, lse = ns["usp_hip_attn_forward"](tq, tk, tv, 0.0, None, causal=causal)
    ro, rl = O.attention_ref(q, k, v, causal=causal)
    assert_close(out.float().cpu().numpy(), ro, *TOL[dt]["out"], "stub out")
    assert_close(lse.cpu().numpy(), rl, 2e-3, 1e-4, "stub lse")
    dq, dk, dv = torch.empty_like(tq), torch.empty_like(tk), torch.empty_like(tv)
    ns["usp_hip_attn_backward"](tdo, tq, tk, tv, out, lse, dq, dk, dv, 0.0, None, causal)
    torch.cuda.synchronize()
    rdq, rdk, rdv = O.block_bwd(do, q, k, v, out.float().cpu().numpy(), rl, None, causal)
    g = Hq // Hkv
    assert_close(dq.float().cpu().numpy(), rdq, *TOL[dt]["grad"], "stub dq")
    assert_close(dk.float().cpu().numpy(), rdk, *grad_tol(dt, g), "stub dk")
    assert_close(dv.float().cpu().numpy(), rdv, *grad_tol(dt, g), "stub dv")


def test_documented_stub_with_a_sliding_window():
    """The stub forwards `window_size` (ABI v5, USP_ATTN_WINDOW): forward and backward against the windowed oracle."""
    ns = _stub_namespace()
    dev = torch.device("cuda:0")
    B, S, H, D, dt, win = 1, 448, 2, 64, "bfloat16", (96, 0)
    rs = np.random.RandomState(8)
    q, k, v, do = (round_to(rs.standard_normal((B, S, H, D)).astype(np.float32), dt) for _ in range(4))
    tq, tk, tv, tdo = (torch.from_numpy(x).to(torch.bfloat16).to(dev) for x in (q, k, v, do))
    out, lse = ns["usp_hip_attn_forward"](tq, tk, tv, 0.0, None, causal=True, window_size=win)
    ro, rl = O.attention_ref(q, k, v, causal=True, window=win)
    assert_close(out.float().cpu().numpy(), ro, *TOL[dt]["out"], "stub out")
    dq, dk, dv = torch.empty_like(tq), torch.empty_like(tk), torch.empty_like(tv)
    ns["usp_hip_attn_backward"](tdo, tq, tk, tv, out, lse, dq, dk, dv, 0.0, None, True, win)
    torch.cuda.synchronize()
    for got, ref, name in zip((dq, dk, dv), O.block_bwd(do, q, k, v, out.float().cpu().numpy(), rl, None, True, window=win),
                              ("dq", "dk", "dv")):
        assert_close(got.float().cpu().numpy(), ref, *TOL[dt]["grad"], f"stub {name}")
