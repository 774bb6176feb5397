"""Golden vectors for the sliding-window mask (round 3): outputs of the REFERENCE's own test/test_utils.py:attention_ref
(the function its hybrid tests hold results against, test_hybrid_attn.py:360-386) with `window_size` set, fp32 upcast,
on seeded inputs -- what oracle/usp_oracle.py's `window` argument is pinned to (tests/test_oracle_golden.py).

    python tests/golden/make_golden_window.py      # needs /root/reference; writes tests/golden/w_window_ref.npz

Cases keep to windows the reference's construct_local_mask defines the way flash-attn does: both bounds given, or no
left bound (a left bound with right = -1 is "unbounded" in flash-attn and "-1" in that helper)."""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True          # /root/reference is read-only: importing from it must not leave a __pycache__ there
sys.path.insert(0, "/root/reference/test")
from test_utils import attention_ref  # noqa: E402

CASES = [  # Sq, Sk, Hq, Hkv, D, causal, (left, right)
    (64, 64, 2, 2, 16, False, (8, 0)),
    (48, 80, 4, 2, 16, True, (16, 0)),
    (80, 48, 2, 1, 16, False, (5, 7)),
    (64, 64, 2, 2, 16, False, (-1, 3)),
    (96, 96, 2, 2, 32, True, (0, 0)),
    (70, 133, 3, 1, 16, False, (40, 12)),
]


def main():
    out = {"n": np.int64(len(CASES))}
    for i, (Sq, Sk, Hq, Hkv, D, causal, win) in enumerate(CASES):
        rs = np.random.RandomState(100 + i)
        q, k, v = (rs.standard_normal((1, S, H, D)).astype(np.float32) for S, H in ((Sq, Hq), (Sk, Hkv), (Sk, Hkv)))
        o, _ = attention_ref(torch.from_numpy(q), torch.from_numpy(k), torch.from_numpy(v), causal=causal,
                             window_size=win, upcast=True)
        out[f"case{i}"] = np.array([Sq, Sk, Hq, Hkv, D, int(causal), win[0], win[1]], dtype=np.int64)
        out[f"out{i}"] = o.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "w_window_ref.npz"), **out)


if __name__ == "__main__":
    main()
