"""DEV TOOL: what ONE rank of a multi-GPU BASELINE config costs on a dedicated MI355X with the wire replaced by local copies.

The real layer (LongContextAttention: packed + pipelined exchange, ring schedule, side streams, interleavable launches,
pack / unpack / add / cast kernels, autograd) runs in ONE process as rank `--rank` of a ulysses x ring grid; every
torch.distributed call of the package is served by a stand-in (all_to_all_single and send/recv = local copies of the same
size, on the same side streams).  The result is the compute-only iteration of that rank, i.e. the per-GPU ceiling of the
config before any link time -- measurable on a 1-GPU box.

    python tools/rank_emulation.py --gpus 8 [--rank 0] [--iters 5] [--env USP_PIPELINE_ULYSSES=0]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


class Group:
    def __init__(self, size, rank):
        self.size, self.rank = size, rank


class _Req:
    def wait(self):
        pass


class FakeDist:
    """torch.distributed as seen by yunchang_amd: groups are Group objects, transfers are local copies."""
    isend, irecv = "send", "recv"
    ProcessGroup = object

    def get_world_size(self, group=None):
        return group.size

    def get_rank(self, group=None):
        return group.rank

    def get_global_rank(self, group, r):
        return r

    def P2POp(self, op, tensor, peer, group=None):
        return (op, tensor)

    def batch_isend_irecv(self, ops):
        sends = [t for o, t in ops if o == "send"]
        for i, (o, t) in enumerate(x for x in ops if x[0] == "recv"):
            src = next((s for s in sends[i % max(1, len(sends)):] + sends if s.shape == t.shape), None)
            if src is not None:
                t.copy_(src)
        return [_Req()]

    def all_to_all_single(self, recv, send, group=None):
        recv.copy_(send)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--env", action="append", default=[])
    args = ap.parse_args()
    for kv in args.env:
        k, v = kv.split("=", 1)
        os.environ[k] = v
    import bench
    import yunchang_amd as Y
    import yunchang_amd.comm.all_to_all as A
    import yunchang_amd.hybrid.async_attn_layer as AL
    import yunchang_amd.hybrid.attn_layer as HL
    import yunchang_amd.ring.ring_flash_attn as RB
    import yunchang_amd.ring.stripe_flash_attn as RS
    import yunchang_amd.ring.utils as U
    import yunchang_amd.ring.zigzag_ring_flash_attn as RZ
    fake = FakeDist()
    for mod in (A, AL, HL, RB, RS, U, RZ):
        mod.dist = fake
    cfg = (bench.CONFIG_WORKLOADS if os.environ.get("USP_BENCH_WORKLOAD") == "configs" else bench.WORKLOADS)[args.gpus]
    ud, rd = cfg["ud"], cfg["rd"]
    u_rank, r_rank = args.rank % ud, args.rank // ud
    Y.PROCESS_GROUP.ULYSSES_PG, Y.PROCESS_GROUP.RING_PG = Group(ud, u_rank), Group(rd, r_rank)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    n = args.gpus
    B, Sl, Hq, Hkv, D = cfg["B"], cfg["S"] // n, cfg["Hq"], cfg["Hkv"], cfg["D"]
    g = torch.Generator(device=dev).manual_seed(0)
    lq, ldo = (torch.randn((B, Sl, Hq, D), device=dev, generator=g).to(torch.bfloat16) for _ in range(2))
    lk, lv = (torch.randn((B, Sl, Hkv, D), device=dev, generator=g).to(torch.bfloat16) for _ in range(2))
    if cfg["bwd"]:
        for t in (lq, lk, lv):
            t.requires_grad_(True)
    attn = Y.LongContextAttention(ring_impl_type=cfg["impl"], attn_type=Y.AttnType.HIP)

    def step():
        out = attn(lq, lk, lv, causal=True)
        if cfg["bwd"]:
            out.backward(ldo)
            lq.grad = lk.grad = lv.grad = None

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.iters * 1e3
    flops = bench.fwd_flops(cfg["B"], cfg["Hq"], cfg["S"], cfg["D"]) * (3.5 if cfg["bwd"] else 1.0) / n
    tf = flops / (ms * 1e-3) / 1e12
    print(f"{cfg['name']}\n  rank {args.rank} (ulysses {u_rank}/{ud}, ring {r_rank}/{rd}), wire = local copies, "
          f"env {args.env or '-'}: {ms:8.3f} ms per iteration = {tf:7.1f} TFLOP/s per GPU = "
          f"{tf / bench.PEAK_BF16_TFLOPS * 100:4.1f} % of the MFMA roofline "
          f"(exchange: {bench.exchange_mode(attn, lq, lk, cfg, n)})")


if __name__ == "__main__":
    main()
