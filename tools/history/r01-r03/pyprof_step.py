import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29761")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
import yunchang_amd as Y
Y.set_seq_parallel_pg(1, 1, 0, 1)
B, S, H, D = 2, 8192, 16, 128
q, k, v = (torch.randn(B, S, H, D, device="cuda").to(torch.bfloat16) for _ in range(3))
attn = Y.LongContextAttention(ring_impl_type="basic", attn_type=Y.AttnType.HIP)
for _ in range(5): attn(q, k, v, causal=True)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n): attn(q, k, v, causal=True)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"CPU issue time per step {1e6*(t1-t0)/n:.1f} us ; wall per step {1e6*(t2-t0)/n:.1f} us")
pr = cProfile.Profile(); pr.enable()
for _ in range(n): attn(q, k, v, causal=True)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
