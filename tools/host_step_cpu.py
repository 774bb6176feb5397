"""Host cost of one N=1 step (LongContextAttention -> ring -> _C.flash_fwd) with the device launch stubbed out:
runs anywhere (no GPU).  usage: python tools/host_step_cpu.py [--profile] [--bwd]"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29763")
dist.init_process_group("gloo", rank=0, world_size=1)
import yunchang_amd as Y
from yunchang_amd import _C


class _Lib:
    def __getattr__(self, name):
        real = getattr(_C_real, name)
        if name.endswith("_bytes") or name in ("usp_strerror", "usp_abi_version"):
            return real
        return lambda *a: 0


_C_real = _C.load()
_C._lib = _Lib()
_C._require_cuda = lambda *t: None
_C._stream = lambda: None
Y.set_seq_parallel_pg(1, 1, 0, 1)
B, S, H, D = 2, 512, 16, 128
bwd = "--bwd" in sys.argv
q, k, v, do = (torch.randn(B, S, H, D).to(torch.bfloat16) for _ in range(4))
if bwd:
    for t in (q, k, v):
        t.requires_grad_(True)
attn = Y.LongContextAttention(ring_impl_type="basic", attn_type=Y.AttnType.HIP)


def step():
    out = attn(q, k, v, causal=True)
    if bwd:
        out.backward(do)
        q.grad = k.grad = v.grad = None


for _ in range(20):
    step()
n = 2000
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
print(f"host time per step {1e6 * (t1 - t0) / n:.1f} us (launch stubbed, {'fwd+bwd' if bwd else 'fwd'})")
if "--profile" in sys.argv:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(n):
        step()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
