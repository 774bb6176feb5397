"""HBM-bound helpers of the Ulysses exchange (usp_copy_rows) at BASELINE shapes: achieved GB/s (dev tool)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import yunchang_amd  # noqa
from yunchang_amd.comm import all_to_all as A
dev = torch.device("cuda:0")


def timeit(f, n=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, (B, Sl, H, D, P) in {"C3 q (S/P=8192,H16)": (1, 8192, 16, 128, 2), "C5 q (S/ws=8192,H32)": (1, 8192, 32, 128, 2),
                               "C5 k (H4)": (1, 8192, 4, 128, 2), "B=2 q": (2, 4096, 16, 128, 2)}.items():
    x = torch.randn(B, Sl, H, D, device=dev, dtype=torch.bfloat16)
    nbytes = x.numel() * 2
    t1 = timeit(lambda: A.pack_heads(x, P))
    recv = A.pack_heads(x, P)
    full = A.view_seq(recv)                      # (B, S, H/P, D) view
    y = torch.randn(B, Sl * P, H // P, D, device=dev, dtype=torch.bfloat16)
    t2 = timeit(lambda: A.pack_seq(y, P))
    r2 = A.pack_seq(y, P)
    t3 = timeit(lambda: A.unpack_heads(r2))
    gb = lambda t: 2 * nbytes / t / 1e6          # read + write
    print(f"PACK {name:24s} {nbytes / 2**20:6.1f} MiB  pack_heads {t1 * 1e3:7.1f} us {gb(t1):7.0f} GB/s | pack_seq {t2 * 1e3:7.1f} us {gb(t2):7.0f} GB/s | unpack_heads {t3 * 1e3:7.1f} us {gb(t3):7.0f} GB/s")
