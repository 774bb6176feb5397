#!/usr/bin/env python3
"""bench.py -- the USP attention hot path on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (LongContextAttention: Ulysses all-to-all x zigzag ring x HIP
flash kernels) over one batch of synthetic N(0,1) bf16 tensors resident in HBM.  Workload per N is
the BASELINE.json configuration for that GPU count (tokens per GPU fixed at 8192 => "weak"):
    N=1  configs[1]  B2 S8192  H16/16 D128 causal fwd          ulysses1 x ring1
    N=2  configs[2]  B1 S16384 H16/16 D128 causal fwd          ulysses2 x ring1
    N=4  configs[3]  B1 S32768 H16/16 D128 causal fwd          ulysses1 x ring4 zigzag
    N=8  configs[4]  B1 S65536 H32/4  D128 causal fwd+bwd      ulysses2 x ring4 zigzag
(B and causal are not stated for configs[2..4]; B=1 and causal=True are assumed)
Rank 0 prints ONE JSON line.  value = whole-job algorithmic TFLOP/s: fwd 4*B*Hq*S^2*D/2 (causal),
bwd 2.5x fwd, no credit for masked tiles, recompute or merges.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

PEAK_BF16_TFLOPS = 2500.0      # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md:42

WORKLOADS = {
    1: dict(name="configs[1]: 1xMI355X ring=1 ulysses=1 B=2 S=8192 H=16 D=128 bf16 causal fwd",
            B=2, S=8192, Hq=16, Hkv=16, D=128, ud=1, rd=1, impl="basic", bwd=False),
    2: dict(name="configs[2]: 2xMI355X ulysses=2 ring=1 B=1 S=16384 H=16 D=128 bf16 causal fwd",
            B=1, S=16384, Hq=16, Hkv=16, D=128, ud=2, rd=1, impl="basic", bwd=False),
    4: dict(name="configs[3]: 4xMI355X ulysses=1 ring=4 zigzag B=1 S=32768 H=16 D=128 bf16 causal fwd",
            B=1, S=32768, Hq=16, Hkv=16, D=128, ud=1, rd=4, impl="zigzag", bwd=False),
    8: dict(name="configs[4]: 8xMI355X ulysses=2 ring=4 zigzag B=1 S=65536 GQA H=32/Hkv=4 D=128 bf16 causal fwd+bwd",
            B=1, S=65536, Hq=32, Hkv=4, D=128, ud=2, rd=4, impl="zigzag", bwd=True),
}


def fwd_flops(B, Hq, S, D, causal=True):
    return 4.0 * B * Hq * S * S * D * (0.5 if causal else 1.0)


def make_global(cfg, dev):
    """Same N(0,1) global tensors on every rank (same seed, same device type): no broadcast needed.
    The reference protocol seeds rank 0 and broadcasts (test/test_hybrid_attn.py:125-184)."""
    g = torch.Generator(device=dev).manual_seed(0)
    sh_q = (cfg["B"], cfg["S"], cfg["Hq"], cfg["D"])
    sh_k = (cfg["B"], cfg["S"], cfg["Hkv"], cfg["D"])
    q = torch.randn(sh_q, device=dev, dtype=torch.float32, generator=g).to(torch.bfloat16)
    k = torch.randn(sh_k, device=dev, dtype=torch.float32, generator=g).to(torch.bfloat16)
    v = torch.randn(sh_k, device=dev, dtype=torch.float32, generator=g).to(torch.bfloat16)
    do = torch.randn(sh_q, device=dev, dtype=torch.float32, generator=g).to(torch.bfloat16)
    return q, k, v, do


def local_row_ranges(cfg, rank, ws):
    """Global row ranges [a, b) this rank owns under the layout (for the in-bench parity check)."""
    S, ud, rd = cfg["S"], cfg["ud"], cfg["rd"]
    if cfg["impl"] == "basic":
        n = S // ws
        return [(rank * n, (rank + 1) * n)]
    r_rank, u_rank = rank // ud, rank % ud            # use_ulysses_low grid (globals.py:39-57)
    c = S // (2 * rd)
    sub = c * 2 // ud                                 # rows per ulysses rank inside [chunk r | chunk 2rd-1-r]
    rows = list(range(r_rank * c, (r_rank + 1) * c)) + list(range((2 * rd - 1 - r_rank) * c, (2 * rd - r_rank) * c))
    mine = rows[u_rank * sub:(u_rank + 1) * sub]
    out, a = [], mine[0]
    for i in range(1, len(mine) + 1):
        if i == len(mine) or mine[i] != mine[i - 1] + 1:
            out.append((a, mine[i - 1] + 1))
            if i < len(mine):
                a = mine[i]
    return out


def parity_check(cfg, rank, ws, out_local, q, k, v):
    """max |USP shard - single-GPU kernel on the same global rows| (outside the timed region)."""
    from yunchang_amd.kernels import hip_attn_forward
    worst, pos = 0.0, 0
    for a, b in local_row_ranges(cfg, rank, ws):
        ref, _ = hip_attn_forward(q[:, a:b], k[:, :b], v[:, :b], causal=True)     # bottom-right causal
        got = out_local[:, pos:pos + (b - a)]
        worst = max(worst, float((got.float() - ref.float()).abs().max()))
        pos += b - a
    return worst


def kernel_roofline(cfg, dev, iters=20):
    """Dominant kernel (flash_fwd_kernel) timed alone, live, with device events on the stream the
    kernel is launched on (torch's current stream)."""
    from yunchang_amd import _C
    B, S, Hq, Hkv, D = cfg["B"], cfg["S"], cfg["Hq"], cfg["Hkv"], cfg["D"]
    g = torch.Generator(device=dev).manual_seed(1)
    q = torch.randn((B, S, Hq, D), device=dev, generator=g).to(torch.bfloat16)
    k = torch.randn((B, S, Hkv, D), device=dev, generator=g).to(torch.bfloat16)
    v = torch.randn((B, S, Hkv, D), device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty_like(q)
    lse = torch.empty((B, Hq, S), device=dev, dtype=torch.float32)
    scale = D ** -0.5
    for _ in range(3):
        _C.flash_fwd(q, k, v, scale, True, lse, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        _C.flash_fwd(q, k, v, scale, True, lse, out)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / iters
    achieved = fwd_flops(B, Hq, S, D) / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "usp::flash_fwd_kernel<128,bf16,causal>", "achieved": round(achieved, 1),
            "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
            "kernel_ms": round(ms, 4), "traffic": None}


def cpu_baseline(cfg):
    """The CPU port (oracle/attn_oracle.c, OpenMP over (batch, head)) and the torch CPU op the
    reference's TORCH_EFFICIENT path falls back to, on a bounded sample of the N=1 workload."""
    cores = os.cpu_count() or 1
    S, D = 2048, cfg["D"]
    H = max(2, min(cores, 16))
    rs = np.random.RandomState(0)
    q, k, v = (rs.standard_normal((1, S, H, D)).astype(np.float32) for _ in range(3))
    fl = fwd_flops(1, H, S, D)
    res = {"unit": "TFLOP/s", "cores": cores, "kind": "port",
           "sample": f"causal fwd B1 S{S} H{H} D{D} fp32 (same per-head problem as the workload, "
                     f"1/16 of its sequence); oracle/attn_oracle.c with OpenMP"}
    so = os.path.join(ROOT, "oracle", "libattn_oracle.so")
    try:
        L = ctypes.CDLL(so)
        out = np.empty_like(q)
        lse = np.empty((1, H, S), np.float32)
        fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        t0 = time.perf_counter()
        L.usp_oracle_attn_fwd(fp(q), fp(k), fp(v), 1, S, S, H, H, D, ctypes.c_float(D ** -0.5), 1, fp(out), fp(lse))
        dt = time.perf_counter() - t0
        res["value"] = round(fl / dt / 1e12, 5)
        res["seconds"] = round(dt, 2)
    except OSError as e:
        res["value"] = None
        res["error"] = str(e)
    try:   # the reference's CPU substitute for TORCH_EFFICIENT, same sample, bf16
        torch.set_num_threads(cores)
        tq, tk, tv = (torch.from_numpy(x).to(torch.bfloat16).transpose(1, 2) for x in (q, k, v))
        op = torch.ops.aten._scaled_dot_product_flash_attention_for_cpu
        op(tq, tk, tv, 0.0, True)
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            op(tq, tk, tv, 0.0, True)
        dt = (time.perf_counter() - t0) / n
        res["torch_cpu_flash_bf16_value"] = round(fl / dt / 1e12, 5)
    except Exception as e:  # pragma: no cover
        res["torch_cpu_flash_bf16_value"] = None
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--bwd", type=int, default=-1, help="override: 1 = fwd+bwd, 0 = fwd only")
    args = ap.parse_args()

    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if ws != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ws}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if args.gpus not in WORKLOADS:
        raise SystemExit(f"--gpus must be one of {sorted(WORKLOADS)}")
    cfg = dict(WORKLOADS[args.gpus])
    if args.bwd >= 0:
        cfg["bwd"] = bool(args.bwd)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if ws == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29751")
    dist.init_process_group("nccl", rank=rank, world_size=ws)

    import yunchang_amd as Y
    Y.set_seq_parallel_pg(cfg["ud"], cfg["rd"], rank, ws)
    q, k, v, do = make_global(cfg, dev)
    ext = Y.EXTRACT_FUNC_DICT[cfg["impl"]]
    lq, lk, lv, ldo = (ext(t, rank, world_size=ws, rd=cfg["rd"], ud=cfg["ud"]).detach().clone()
                       for t in (q, k, v, do))
    if cfg["bwd"]:
        for t in (lq, lk, lv):
            t.requires_grad_(True)
    attn = Y.LongContextAttention(ring_impl_type=cfg["impl"], attn_type=Y.AttnType.HIP)

    def step():
        out = attn(lq, lk, lv, causal=True)
        if cfg["bwd"]:
            out.backward(ldo)
            lq.grad = lk.grad = lv.grad = None
        return out

    parity = None
    out = step()
    if not args.no_parity:
        parity = parity_check(cfg, rank, ws, out.detach(), q, k, v)
        pt = torch.tensor([parity], device=dev)
        dist.all_reduce(pt, op=dist.ReduceOp.MAX)
        parity = float(pt.item())
    del q, k, v, do, out
    for _ in range(args.warmup):
        step()

    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    ms = dt / args.steps * 1e3
    flops = fwd_flops(cfg["B"], cfg["Hq"], cfg["S"], cfg["D"]) * (3.5 if cfg["bwd"] else 1.0)
    value = flops / (ms * 1e-3) / 1e12

    if rank == 0:
        line = {
            "metric": "attention TFLOP/s (algorithmic, causal) of LongContextAttention ulysses x ring",
            "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": ws, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": cfg["name"], "global_shape_BSHD": [cfg["B"], cfg["S"], cfg["Hq"], cfg["D"]],
                       "kv_heads": cfg["Hkv"], "parallelism": f"ulysses{cfg['ud']}xring{cfg['rd']}",
                       "layout": cfg["impl"], "pass": "fwd+bwd" if cfg["bwd"] else "fwd",
                       "tokens_per_gpu": cfg["S"] * cfg["B"] // ws,
                       "assumed": "B=1 and causal=True where BASELINE.json's config string is silent"},
            "frac_of_mfma_roofline": round(value / (ws * PEAK_BF16_TFLOPS), 4),
            "parity_max_abs_err_vs_single_gpu_kernel": parity,
        }
        if ws == 1:
            line["roofline"] = kernel_roofline(cfg, dev)
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(line), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
