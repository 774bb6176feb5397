"""Python-side cost of one step of the N > 1 schedules (DEV TOOL, no GPU): the real layer on gloo ranks with every
kernel skipped and every transfer dropped, i.e. what the host spends walking the schedule -- the time the GPU queue must
be deep enough to hide.  USP_HOST_COST_GC=freeze applies bench.py's gc.collect() + gc.freeze() first (a full collection
of torch's ~170 k objects takes 40-60 ms and lands in a 50-step loop now and then)."""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
from dist_util import run_distributed

def worker(rank, ws, ud, rd, bwd, pieces):
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    import yunchang_amd as Y
    import yunchang_amd.ring.utils as U
    import yunchang_amd.comm.all_to_all as A
    import yunchang_amd.hybrid.async_attn_layer as AL
    from yunchang_amd.kernels import set_block_backend
    set_block_backend(b._NoCompute())
    if pieces: os.environ["USP_ZZ_PIECES"] = str(pieces)
    AL._FILL_ITEMS = 1
    Y.set_seq_parallel_pg(ud, rd, rank, ws)
    U.RingComm.commit = lambda self: setattr(self, "_reqs", [])
    A._exchange = lambda send, group, use_sync: send
    B, S, Hq, Hkv, D = 1, 64 * ws, 8, (8 if not bwd else 4), 32
    lq = torch.randn(B, S // ws, Hq, D).to(torch.bfloat16).requires_grad_(bwd)
    lk = torch.randn(B, S // ws, Hkv, D).to(torch.bfloat16).requires_grad_(bwd)
    lv = torch.randn(B, S // ws, Hkv, D).to(torch.bfloat16).requires_grad_(bwd)
    do = torch.randn(B, S // ws, Hq, D).to(torch.bfloat16)
    attn = Y.LongContextAttention(ring_impl_type="zigzag" if rd > 1 else "basic")
    def step():
        o = attn(lq, lk, lv, causal=True)
        if bwd:
            o.backward(do); lq.grad = lk.grad = lv.grad = None
    for _ in range(5): step()
    if os.environ.get('USP_HOST_COST_GC') == 'freeze':
        gc.collect(); gc.freeze()
    t0 = time.perf_counter(); n = 50
    for _ in range(n): step()
    return (time.perf_counter() - t0) / n * 1e3

if __name__ == "__main__":
    for ws, ud, rd, bwd, pieces in ((2, 2, 1, False, 0), (4, 1, 4, False, 1), (4, 1, 4, False, 2), (8, 2, 4, True, 0)):
        r = run_distributed(worker, ws, ud, rd, bwd, pieces)
        print(f"ws {ws} ud {ud} rd {rd} bwd {bwd} pieces {pieces}: host ms per iteration (python only, no kernels / transport): per rank {[round(x,3) for x in r]}")
