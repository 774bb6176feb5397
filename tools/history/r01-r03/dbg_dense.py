"""Stress: dense forward on a ragged shape, repeated launches (dev tool).  usage: dbg_dense.py [lib]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import yunchang_amd
from yunchang_amd import _C
from oracle import usp_oracle as O
if len(sys.argv) > 1:
    _C._LIB_PATH = os.path.abspath(sys.argv[1]); _C.ABI_VERSION = None
    import ctypes
    L = ctypes.CDLL(_C._LIB_PATH); L.usp_abi_version.restype = ctypes.c_int; _C.ABI_VERSION = L.usp_abi_version()
_C.load()
dev = torch.device("cuda:0")


def case(B, S, H, D, reps):
    rs = np.random.RandomState(30)
    tq, tk, tv = (torch.from_numpy(rs.standard_normal((B, S, H, D)).astype(np.float32)).to(torch.bfloat16).to(dev) for _ in range(3))
    q, k, v = (t.float().cpu().numpy() for t in (tq, tk, tv))
    ro, rl = O.attention_ref(q, k, v, True)
    nbad, first_bad = 0, None
    for rep in range(reps):
        out = torch.full((B, S, H, D), float("nan"), dtype=torch.bfloat16, device=dev)
        lse = torch.full((B, H, S), float("nan"), dtype=torch.float32, device=dev)
        _C.flash_fwd(tq, tk, tv, D ** -0.5, True, lse, out=out)
        torch.cuda.synchronize()
        err = np.abs(out.float().cpu().numpy() - ro)
        bad = ~(err <= 0.02 + 0.02 * np.abs(ro))
        if bad.any():
            nbad += 1
            if first_bad is None:
                idx = np.argwhere(bad)[0]
                first_bad = (rep, idx.tolist(), float(err[tuple(idx)]))
    print(f"DENSE-STRESS B{B} S{S} H{H} D{D}: {nbad}/{reps} runs with errors, first {first_bad}", flush=True)


case(1, 130, 4, 128, 30)
case(1, 128, 4, 128, 30)
case(2, 200, 3, 128, 30)
case(1, 1000, 8, 128, 20)
case(1, 1024, 8, 64, 20)
