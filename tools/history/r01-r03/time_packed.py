"""Times the packed variable-length kernels against the dense kernels on the same token count (dev tool)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import yunchang_amd  # noqa
from yunchang_amd import _C
_C.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
H, Hkv, D = 16, 16, 128


def timeit(f, iters=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def run(tag, lens):
    T = sum(lens)
    q = torch.randn(T, H, D, device=dev, dtype=torch.bfloat16)
    k = torch.randn(T, Hkv, D, device=dev, dtype=torch.bfloat16)
    v = torch.randn(T, Hkv, D, device=dev, dtype=torch.bfloat16)
    do = torch.randn(T, H, D, device=dev, dtype=torch.bfloat16)
    out = torch.empty_like(q)
    lse = torch.empty(H, T, device=dev, dtype=torch.float32)
    delta = torch.empty(H, T, device=dev, dtype=torch.float32)
    first = np.concatenate([[0], np.cumsum(lens)[:-1]])
    tb = torch.tensor(np.stack([first, lens], 1), dtype=torch.int32, device=dev)
    mx = max(lens)
    fwd = lambda: _C.flash_fwd_packed(q, k, v, tb, tb, mx, mx, D ** -0.5, True, lse, out=out)
    fwd()
    _C.bwd_delta(do[None], out[None], delta[None])
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    bwd = lambda: _C.flash_bwd_packed(do, q, k, v, lse, delta, tb, tb, mx, mx, None, None, None, D ** -0.5, True,
                                      dq16=dq, dk16=dk, dv16=dv)
    fl = sum(4.0 * H * n * n * D * 0.5 for n in lens)
    tf, tb_ = timeit(fwd), timeit(bwd)
    print(f"PACKED {tag:38s} fwd {tf:7.4f} ms {fl / tf / 1e9:7.1f} TF/s | bwd {tb_:7.4f} ms {2.5 * fl / tb_ / 1e9:7.1f} TF/s")


run("1 x 16384", [16384])
run("2 x 8192", [8192, 8192])
run("8 x 2048", [2048] * 8)
run("32 x 512", [512] * 32)
run("mixed 8192+4096+2048+1024+512x2", [8192, 4096, 2048, 1024, 512, 512])
run("ragged 5000+3000+777+8000", [5000, 3000, 777, 8000])
# dense reference for the first two
for B, S in ((1, 16384), (2, 8192), (8, 2048)):
    q = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
    k, v = torch.randn_like(q), torch.randn_like(q)
    out = torch.empty_like(q)
    lse = torch.empty(B, H, S, device=dev, dtype=torch.float32)
    t = timeit(lambda: _C.flash_fwd(q, k, v, D ** -0.5, True, lse, out=out))
    fl = 4.0 * B * H * S * S * D * 0.5
    print(f"DENSE  B{B} S{S}  fwd {t:7.4f} ms {fl / t / 1e9:7.1f} TF/s")
