"""Can another stream's kernel start while a flash kernel runs?  (dev tool; the mechanism behind
USP_LAUNCH_INTERLEAVE).  A small elementwise kernel stands in for RCCL's send/recv kernels: it is queued on a
side stream right after the attention launch; we report when it finishes relative to the attention kernel."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import yunchang_amd  # noqa
from yunchang_amd import _C
_C.load()
dev = torch.device("cuda:0")
B, Sq, Sk, Hq, Hkv, D = 1, 16384, 8192, 16, 2, 128      # a C5 ring step
q = torch.randn(B, Sq, Hq, D, device=dev, dtype=torch.bfloat16)
k = torch.randn(B, Sk, Hkv, D, device=dev, dtype=torch.bfloat16)
v = torch.randn_like(k)
out = torch.empty_like(q); lse = torch.empty(B, Hq, Sq, device=dev, dtype=torch.float32)
x = torch.randn(16 * 1024 * 1024, device=dev)           # 64 MiB "payload"
y = torch.empty_like(x)
side = torch.cuda.Stream()
for interleave in (False, True):
    for _ in range(3):
        _C.flash_fwd(q, k, v, D ** -0.5, False, lse, out, interleave=interleave)
    torch.cuda.synchronize()
    res = []
    for rep in range(5):
        s0, a1, c1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        s0.record()
        _C.flash_fwd(q, k, v, D ** -0.5, False, lse, out, interleave=interleave)
        a1.record()
        with torch.cuda.stream(side):
            side.wait_event(s0)
            torch.add(x, 1.0, out=y)                     # ~25 us alone
            c1.record(side)
        torch.cuda.synchronize()
        res.append((s0.elapsed_time(a1), s0.elapsed_time(c1)))
    att = sum(r[0] for r in res) / len(res); cp = sum(r[1] for r in res) / len(res)
    print(f"OVERLAP interleave={interleave}: attention {att:.3f} ms, side-stream kernel done at {cp:.3f} ms after start")
