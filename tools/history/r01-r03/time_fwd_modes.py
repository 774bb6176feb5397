"""Times usp_flash_fwd in its three epilogue modes on a ring-step shaped block (dev tool).
usage: python tools/time_fwd_modes.py [libpath]   (libpath: alternative libusp_hip.so for A/B runs)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import yunchang_amd  # noqa
from yunchang_amd import _C
if len(sys.argv) > 1:
    _C._LIB_PATH = os.path.abspath(sys.argv[1])
_C.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)


def run(tag, B, Sq, Sk, Hq, Hkv, D, causal, merge, fe_frac, iters=30):
    q = torch.randn(B, Sq, Hq, D, device=dev, dtype=torch.bfloat16)
    k = torch.randn(B, Sk, Hkv, D, device=dev, dtype=torch.bfloat16)
    v = torch.randn(B, Sk, Hkv, D, device=dev, dtype=torch.bfloat16)
    out = torch.empty_like(q)
    acc = torch.randn(B, Sq, Hq, D, device=dev, dtype=torch.float32)
    lse = torch.randn(B, Hq, Sq, device=dev, dtype=torch.float32)
    fe = int(Sq * fe_frac)
    f = lambda: _C.flash_fwd(q, k, v, D ** -0.5, causal, lse, out, acc, merge, 0, fe)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 4.0 * B * Hq * Sq * Sk * D * (0.5 if causal else 1.0)
    print(f"MODE {tag:34s} {ms:8.4f} ms  {fl / ms / 1e9:8.1f} TFLOP/s")


S = (1, 16384, 8192, 16, 2, 128)            # C5 ring step: q 2c=16384 rows x k c=8192 (GQA 16/2)
run("adopt, all rows final (16-bit)", *S, False, False, 1.0)
run("adopt, none final (fp32 acc)", *S, False, False, 0.0)
run("merge, none final (fp32 acc)", *S, False, True, 0.0)
run("merge, all final (16-bit)", *S, False, True, 1.0)
run("merge, first half final", *S, False, True, 0.5)
run("C2 causal adopt final", 2, 8192, 8192, 16, 16, 128, True, False, 1.0)
