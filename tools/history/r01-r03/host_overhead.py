"""DEV TOOL: where does the host time of one N=1 step go?  Enqueue-only timings (no device sync inside the loop)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29755")
torch.cuda.set_device(0); dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1)
import yunchang_amd as Y
from yunchang_amd import _C
from yunchang_amd.kernels import hip_attn_forward
Y.set_seq_parallel_pg(1, 1, 0, 1)
B, S, H, D = 2, 8192, 16, 128
q, k, v = (torch.randn(B, S, H, D, device=dev).to(torch.bfloat16) for _ in range(3))
out = torch.empty_like(q); lse = torch.empty(B, H, S, device=dev, dtype=torch.float32)
attn = Y.LongContextAttention(ring_impl_type="basic", attn_type=Y.AttnType.HIP)
cases = {"_C.flash_fwd (ctypes only)": lambda: _C.flash_fwd(q, k, v, D ** -0.5, True, lse, out),
         "hip_attn_forward (+2 allocs)": lambda: hip_attn_forward(q, k, v, causal=True),
         "LongContextAttention.forward": lambda: attn(q, k, v, causal=True)}
for name, fn in cases.items():
    for _ in range(20): fn()
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    print(f"{name:32s} host enqueue {t_host * 1e6:7.1f} us/step   with device {t_all * 1e6:7.1f} us/step")

# fixed cost of a timed region: t(K) = a + b K  (bench.timed brackets K steps with synchronize on both sides)
fn = cases["LongContextAttention.forward"]
for K in (5, 10, 20, 50, 100, 200):
    ts = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K): fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    print(f"K={K:4d}: {t * 1e3:8.3f} ms total  {t / K * 1e6:7.1f} us/step   (min of 5)")
