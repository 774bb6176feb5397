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
            "kernel_ms": round(ms, 4), "traffic": pmc_traffic()}


def reference_kernel(cfg, dev, ours_tflops, iters=10):
    """The third-party op behind the reference's TORCH_EFFICIENT attention (yunchang/kernels/attention.py:76-86:
    aten::_scaled_dot_product_efficient_attention on (B,H,S,D) views), timed on this GPU on the same workload,
    equal heads (the op has no GQA).  Reported beside our kernel; never part of `value`.  None if the op does
    not run on this box."""
    try:
        B, S, Hq, D = cfg["B"], cfg["S"], cfg["Hq"], cfg["D"]
        g = torch.Generator(device=dev).manual_seed(1)
        q, k, v = (torch.randn((B, S, Hq, D), device=dev, generator=g).to(torch.bfloat16).transpose(1, 2)
                   for _ in range(3))
        op = torch.ops.aten._scaled_dot_product_efficient_attention
        f = lambda: op(q, k, v, None, True, 0.0, True, scale=D ** -0.5)
        for _ in range(2):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            f()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / iters
        tf = fwd_flops(B, Hq, S, D) / (ms * 1e-3) / 1e12
        return {"op": "aten::_scaled_dot_product_efficient_attention (yunchang AttnType.TORCH_EFFICIENT, "
                      "kernels/attention.py:76-86)", "value": round(tf, 1), "unit": "TFLOP/s",
                "kernel_ms": round(ms, 4), "our_kernel_speedup": round(ours_tflops / tf, 2)}
    except Exception as e:                                  # informative only
        return {"op": "aten::_scaled_dot_product_efficient_attention", "value": None, "error": repr(e)[:200]}


def reference_fwdbwd(cfg, dev, ours_tflops, iters=5):
    """torch's own scaled_dot_product_attention (the flash / efficient kernels a ROCm PyTorch ships) forward +
    backward through autograd on the same workload and GPU: what a user of the reference gets on this box
    without flash-attn.  Informative; None if it does not run."""
    try:
        import torch.nn.functional as F
        B, S, Hq, D = cfg["B"], cfg["S"], cfg["Hq"], cfg["D"]
        g = torch.Generator(device=dev).manual_seed(1)
        q, k, v, do = (torch.randn((B, Hq, S, D), device=dev, generator=g).to(torch.bfloat16) for _ in range(4))
        for t in (q, k, v):
            t.requires_grad_(True)

        def step():
            F.scaled_dot_product_attention(q, k, v, is_causal=True).backward(do)
            q.grad = k.grad = v.grad = None
        for _ in range(2):
            step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            step()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / iters
        tf = 3.5 * fwd_flops(B, Hq, S, D) / (ms * 1e-3) / 1e12
        return {"op": "torch.nn.functional.scaled_dot_product_attention fwd+bwd (autograd)", "value": round(tf, 1),
                "unit": "TFLOP/s", "ms": round(ms, 4), "our_step_speedup": round(ours_tflops / tf, 2)}
    except Exception as e:
        return {"op": "torch sdpa fwd+bwd", "value": None, "error": repr(e)[:200]}


def pmc_traffic():
    """HBM bytes per launch of the forward kernel from the committed rocprofv3 PMC passes
    (profiles/r01_rocprof_summary.txt: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate
    passes, C2 shape).  Counters cannot be collected inside this process; null if the file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_rocprof_summary.txt")
    try:
        rd = wr = None
        for ln in open(path):
            if "HBM read bytes/launch" in ln and rd is None:
                rd = float(ln.split("=")[1].split("MB")[0])
            if "HBM write bytes/launch" in ln and wr is None:
                wr = float(ln.split("=")[1].split("MB")[0])
        if rd is None or wr is None:
            return None
        return {"read_MB": rd, "write_MB": wr, "algorithmic_MB": 268.4,
                "source": "profiles/r01_rocprof_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"}
    except OSError:
        return None


def cpu_baseline(cfg):
    """The CPU port (oracle/attn_oracle.c, OpenMP over (batch, head)) and the torch CPU op the
    reference's TORCH_EFFICIENT path falls back to, on a bounded sample of the N=1 workload."""
    cores = os.cpu_count() or 1
    D = cfg["D"]
    # bounded sample: the workload's own sequence length, as many (batch, head) problems as there are
    # cores to run them side by side, capped at the workload's 32 -- about 10-30 s of CPU work
    S = cfg["S"]
    H = max(1, min(cores, cfg["Hq"] * cfg["B"]))
    rs = np.random.RandomState(0)
    q, k, v = (rs.standard_normal((1, S, H, D)).astype(np.float32) for _ in range(3))
    fl = fwd_flops(1, H, S, D)
    res = {"unit": "TFLOP/s", "cores": cores, "threads_used": H, "kind": "port",
           "sample": f"causal fwd, {H} of the workload's {cfg['Hq'] * cfg['B']} (batch, head) problems at its full "
                     f"S={S}, D={D}, fp32 in / fp64 accumulate; oracle/attn_oracle.c, OpenMP over heads"}
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


def barrier(ws):
    if ws > 1:            # a barrier over one rank is empty; NCCL would still launch an all-reduce for it
        dist.barrier()


def _sync(dev):
    if dev.type == "cuda":
        torch.cuda.synchronize()


def timed(step, steps, ws, dev):
    """K steps bracketed by barrier + synchronize on both sides; max over ranks; seconds per step."""
    _sync(dev)
    barrier(ws)
    _sync(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    _sync(dev)
    barrier(ws)
    _sync(dev)
    dt = time.perf_counter() - t0
    if ws > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    return dt / steps


class _NoCompute:
    """comm-only run: every kernel of the block backend is skipped (buffers stay uninitialised)."""
    name = "none"

    def beside_transfers(self):
        return self

    def __getattr__(self, _):
        return lambda *a, **k: None


def overlap_probe(step, steps, ws, dev, t_iter):
    """overlap = 1 - (t_iter - t_compute_only) / t_comm_only   (SURVEY.md section 8d)."""
    import yunchang_amd.comm.all_to_all as A
    import yunchang_amd.ring.utils as U
    from yunchang_amd.kernels import set_block_backend
    # compute-only: same schedule, the wire replaced by local buffers
    ex, commit, wait = A._exchange, U.RingComm.commit, U.RingComm.wait
    A._exchange = lambda send, group, use_sync: send

    def local_commit(self):
        for snd, rcv in zip(self._ops[0::2], self._ops[1::2]):
            rcv.tensor.copy_(snd.tensor)
        self._reqs = []
    U.RingComm.commit = local_commit
    try:
        for _ in range(2):
            step()
        t_comp = timed(step, steps, ws, dev)
    finally:
        A._exchange, U.RingComm.commit, U.RingComm.wait = ex, commit, wait
    # comm-only: same schedule, every kernel skipped
    prev = set_block_backend(_NoCompute())
    try:
        for _ in range(2):
            step()
        t_comm = timed(step, steps, ws, dev)
    finally:
        set_block_backend(prev)
    ov = 1.0 - (t_iter - t_comp) / t_comm if t_comm > 0 else None
    return {"definition": "1 - (t_iter - t_compute_only) / t_comm_only", "value": None if ov is None else round(ov, 4),
            "ms_iter": round(t_iter * 1e3, 4), "ms_compute_only": round(t_comp * 1e3, 4),
            "ms_comm_only": round(t_comm * 1e3, 4),
            "note": "comm-only skips every kernel (incl. pack/unpack); compute-only replaces each transfer by a "
                    "local copy of the same size"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="skip the compute-only / comm-only runs (N > 1)")
    ap.add_argument("--bwd", type=int, default=-1, help="override: 1 = fwd+bwd, 0 = fwd only")
    ap.add_argument("--async-ulysses", action="store_true",
                    help="use AsyncLongContextAttention (head-group pipelined all-to-all) instead of LongContextAttention")
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
    # USP_BENCH_BACKEND=gloo is a DEVELOPMENT smoke mode, not a measurement: all ranks share cuda:0 and
    # talk over gloo, so the N > 1 code path of this script can be exercised on a 1-GPU box (RCCL refuses
    # two ranks on one device).  The line it prints is tagged "smoke" and must not be read as a result.
    backend = os.environ.get("USP_BENCH_BACKEND", "nccl")
    smoke = backend != "nccl"
    if smoke:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if ws == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29751")
    dist.init_process_group(backend, rank=rank, world_size=ws)

    import yunchang_amd as Y
    if smoke:      # gloo's p2p is not stream-ordered for device tensors (tests/test_gpu_multiproc.py)
        import yunchang_amd.ring.utils as _U
        _commit = _U.RingComm.commit
        _U.RingComm.commit = lambda self: (torch.cuda.synchronize(), _commit(self))[1]
    Y.set_seq_parallel_pg(cfg["ud"], cfg["rd"], rank, ws)
    q, k, v, do = make_global(cfg, dev)
    ext = Y.EXTRACT_FUNC_DICT[cfg["impl"]]
    lq, lk, lv, ldo = (ext(t, rank, world_size=ws, rd=cfg["rd"], ud=cfg["ud"]).detach().clone()
                       for t in (q, k, v, do))
    if cfg["bwd"]:
        for t in (lq, lk, lv):
            t.requires_grad_(True)
    if args.async_ulysses:
        attn = Y.AsyncLongContextAttention(ring_impl_type=cfg["impl"])
    else:
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
        try:
            parity = parity_check(cfg, rank, ws, out.detach(), q, k, v)
        except Exception as e:                      # never let the optional check kill the measurement
            print(f"[rank {rank}] parity check failed to run: {e!r}", file=sys.stderr)
            parity = float("nan")
        if ws > 1:
            pt = torch.tensor([parity], device=dev)
            dist.all_reduce(pt, op=dist.ReduceOp.MAX)
            parity = float(pt.item())
    del q, k, v, do, out
    for _ in range(args.warmup):
        step()

    dt = timed(step, args.steps, ws, dev) * args.steps

    ms = dt / args.steps * 1e3
    flops = fwd_flops(cfg["B"], cfg["Hq"], cfg["S"], cfg["D"]) * (3.5 if cfg["bwd"] else 1.0)
    value = flops / (ms * 1e-3) / 1e12

    overlap = None
    if ws > 1 and not args.no_overlap:
        try:
            overlap = overlap_probe(step, max(3, args.steps // 4), ws, dev, ms * 1e-3)
        except Exception as e:
            print(f"[rank {rank}] overlap probe failed to run: {e!r}", file=sys.stderr)
            overlap = {"value": None, "error": repr(e)}

    if rank == 0:
        line = {
            "metric": "attention TFLOP/s (algorithmic, causal) of LongContextAttention ulysses x ring",
            "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": ws, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": cfg["name"], "global_shape_BSHD": [cfg["B"], cfg["S"], cfg["Hq"], cfg["D"]],
                       "kv_heads": cfg["Hkv"], "parallelism": f"ulysses{cfg['ud']}xring{cfg['rd']}",
                       "layout": cfg["impl"], "pass": "fwd+bwd" if cfg["bwd"] else "fwd",
                       "layer": "AsyncLongContextAttention" if args.async_ulysses else "LongContextAttention",
                       "ulysses_exchange": ("pipelined over head groups" if args.async_ulysses or (
                           hasattr(attn, "_pipelined_exchange") and attn._pipelined_exchange(lq, lk)) else "sequential"),
                       "tokens_per_gpu": cfg["S"] * cfg["B"] // ws,
                       "assumed": "B=1 and causal=True where BASELINE.json's config string is silent"},
            "frac_of_mfma_roofline": round(value / (ws * PEAK_BF16_TFLOPS), 4),
            "parity_max_abs_err_vs_single_gpu_kernel": parity,
        }
        if overlap is not None:
            line["overlap"] = overlap
        if smoke:
            line["smoke"] = f"backend={backend}, all ranks on cuda:0 -- NOT a measurement"
        if ws == 1:
            line["roofline"] = kernel_roofline(cfg, dev)
            line["reference_kernel_on_this_gpu"] = reference_kernel(cfg, dev, line["roofline"]["achieved"])
            if cfg["bwd"]:
                line["reference_fwdbwd_on_this_gpu"] = reference_fwdbwd(cfg, dev, value)
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(line), flush=True)
    barrier(ws)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
