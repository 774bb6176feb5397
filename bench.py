#!/usr/bin/env python3
"""bench.py -- the USP attention hot path on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (LongContextAttention: Ulysses all-to-all x zigzag ring x HIP
flash kernels), forward + backward through autograd, over one batch of synthetic N(0,1) bf16 tensors
resident in HBM.  The workload is the configuration BASELINE.json's metric is quoted on -- "attention
TFLOP/s + iter ms at seqlen 64K, ulysses x ring on 1/2/4/8 MI355X" = configs[4]'s global tensors (B1 S65536
GQA H32/Hkv4 D128 bf16 causal fwd+bwd), which fit one MI355X -- on the process grid BASELINE names for the
GPU count (the same global problem at every N => "strong"):
    N=1  ulysses1 x ring1            N=2  ulysses2 x ring1 (configs[2]'s grid)
    N=4  ulysses1 x ring4 zigzag (configs[3]'s grid)           N=8  ulysses2 x ring4 zigzag (configs[4])
(B is not stated for configs[4]; B=1 is assumed)  BASELINE's other configs (C2: B2 S8192 H16
forward on one GPU, C3, C4) are parity-test cases (tests/), and C2 stays in the N = 1 line as a secondary block
(`roofline.c2`).  `USP_BENCH_WORKLOAD=configs` restores the per-N BASELINE configs of rounds 1-4 for A/B runs.
Rank 0 prints ONE JSON line.  value = whole-job algorithmic TFLOP/s: fwd 4*B*Hq*S^2*D/2 (causal),
bwd 2.5x fwd, no credit for masked tiles, recompute or merges; ms_per_step = the iteration time.
"""
import argparse
import ctypes
import gc
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

PEAK_BF16_TFLOPS = 2500.0      # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md:42

_M = dict(B=1, S=65536, Hq=32, Hkv=4, D=128, bwd=True)       # the metric's configuration: configs[4]'s global tensors
_MN = "B=1 S=65536 GQA H=32/Hkv=4 D=128 bf16 causal fwd+bwd"
WORKLOADS = {
    1: dict(_M, name=f"configs[4] global shape on 1xMI355X ring=1 ulysses=1: {_MN}", ud=1, rd=1, impl="zigzag"),
    2: dict(_M, name=f"configs[4] global shape on 2xMI355X ulysses=2 ring=1 (configs[2]'s grid): {_MN}", ud=2, rd=1, impl="basic"),
    4: dict(_M, name=f"configs[4] global shape on 4xMI355X ulysses=1 ring=4 zigzag (configs[3]'s grid): {_MN}", ud=1, rd=4,
            impl="zigzag"),
    8: dict(_M, name=f"configs[4]: 8xMI355X ulysses=2 ring=4 zigzag {_MN}", ud=2, rd=4, impl="zigzag"),
}
# BASELINE.json's per-GPU-count configs (the bench workloads of rounds 1-4; USP_BENCH_WORKLOAD=configs): parity-test cases
CONFIG_WORKLOADS = {
    1: dict(name="configs[1]: 1xMI355X ring=1 ulysses=1 B=2 S=8192 H=16 D=128 bf16 causal fwd",
            B=2, S=8192, Hq=16, Hkv=16, D=128, ud=1, rd=1, impl="basic", bwd=False),
    2: dict(name="configs[2]: 2xMI355X ulysses=2 ring=1 B=1 S=16384 H=16 D=128 bf16 causal fwd",
            B=1, S=16384, Hq=16, Hkv=16, D=128, ud=2, rd=1, impl="basic", bwd=False),
    4: dict(name="configs[3]: 4xMI355X ulysses=1 ring=4 zigzag B=1 S=32768 H=16 D=128 bf16 causal fwd",
            B=1, S=32768, Hq=16, Hkv=16, D=128, ud=1, rd=4, impl="zigzag", bwd=False),
    8: dict(WORKLOADS[8]),
}
C2 = CONFIG_WORKLOADS[1]


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


def nanmax(a, b):
    """max(a, b) that PROPAGATES NaN.  Python's max(0.0, nan) returns 0.0 (every comparison with NaN is False), which
    would silently drop a NaN in a compared row from every parity figure this file prints; a NaN anywhere makes the
    figure NaN, and the tests that read the figure fail on it."""
    return float("nan") if (a != a or b != b) else max(a, b)


def parity_check(cfg, rank, ws, out_local, q, k, v, n_rows=6):
    """The USP shard of this rank against (a) the third-party op behind the reference's AttnType.TORCH_EFFICIENT
    (aten::_scaled_dot_product_efficient_attention, yunchang/kernels/attention.py:76-86; equal heads only, so K/V
    heads are expanded for GQA) on the same global tensors, and (b) exact attention in fp64 for a few sampled rows
    of every owned row range.  Outside the timed region.  Returns (max_abs_err_vs_reference_op | None,
    max_abs_err_vs_fp64_rows)."""
    B, S, Hq, D = q.shape
    g = Hq // k.shape[2]
    scale = D ** -0.5
    ranges = local_row_ranges(cfg, rank, ws)
    worst_op = None
    try:
        op = torch.ops.aten._scaled_dot_product_efficient_attention
        hi = max(b for _, b in ranges)                       # causal: rows < hi only need keys < hi
        kk, vv = (t[:, :hi].repeat_interleave(g, dim=2) if g > 1 else t[:, :hi] for t in (k, v))
        ref = op(q[:, :hi].transpose(1, 2), kk.transpose(1, 2), vv.transpose(1, 2), None, False, 0.0, True,
                 scale=scale)[0].transpose(1, 2)
        worst_op, pos = 0.0, 0
        for a, b in ranges:
            got = out_local[:, pos:pos + (b - a)]
            worst_op = nanmax(worst_op, float((got.float() - ref[:, a:b].float()).abs().max()))
            pos += b - a
        del ref, kk, vv
    except Exception as e:                                   # the op may be missing from a torch build
        print(f"[rank {rank}] reference op not available for the parity check: {repr(e)[:160]}", file=sys.stderr)
    worst_rows, pos = 0.0, 0
    Hkv = k.shape[2]
    for a, b in ranges:
        for row in sorted({a, b - 1, *np.random.RandomState(a).randint(a, b, size=n_rows).tolist()}):
            qd = q[:, row].double().reshape(B, Hkv, g, D)                             # (B,Hkv,g,D): no K/V head expansion
            kd, vd = k[:, :row + 1].double(), v[:, :row + 1].double()                 # (B,row+1,Hkv,D)
            p = torch.softmax(torch.einsum("bkgd,bskd->bkgs", qd, kd) * scale, dim=-1)
            ref_row = torch.einsum("bkgs,bskd->bkgd", p, vd).reshape(B, Hq, D)
            got = out_local[:, pos + row - a].double()
            worst_rows = nanmax(worst_rows, float((got - ref_row).abs().max()))
        pos += b - a
    return worst_op, worst_rows


def _time_events(fn, iters, warm=2):
    """Average device time of fn() in ms: device events on torch's current stream, which is the stream every
    kernel of the package is launched on (_C._stream)."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def _kernel_inputs(B, S, Hq, Hkv, D, dev, seed=1):
    g = torch.Generator(device=dev).manual_seed(seed)
    q, do = (torch.randn((B, S, Hq, D), device=dev, generator=g).to(torch.bfloat16) for _ in range(2))
    k, v = (torch.randn((B, S, Hkv, D), device=dev, generator=g).to(torch.bfloat16) for _ in range(2))
    return q, k, v, do


def _fwd_bwd_kernels(B, S, Hq, Hkv, D, dev, iters, keep=False, family=None, per_kernel=False):
    """Forward kernel and the backward's kernels (delta + dK/dV [+ head reduce] + dQ) timed alone through the
    C ABI on N(0,1) data, device events on the launch stream; algorithmic TFLOP/s (forward 4 B Hq S^2 D / 2, backward
    2.5x).  `family`: "row64" | "wave32" pins the kernel family of every launch (ABI v6), None = the library's dispatch.
    `per_kernel`: also time the backward's launches one by one (delta; dK/dV + reduce; dQ -- USP_BWD_SKIP_*).
    `keep`: also return the inputs and results (for sampled_parity)."""
    from yunchang_amd import _C
    q, k, v, do = _kernel_inputs(B, S, Hq, Hkv, D, dev)
    out = torch.empty_like(q)
    lse = torch.empty((B, Hq, S), device=dev, dtype=torch.float32)
    delta = torch.empty_like(lse)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    scale = D ** -0.5
    kinds = {}
    fwd = lambda: _C.flash_fwd(q, k, v, scale, True, lse, out, family=family)
    bwd_args = (do, q, k, v, lse, delta, None, None, None, scale, True)
    bwd_kw = dict(dq16=dq, dk16=dk, dv16=dv, family=family)

    def bwd():
        _C.bwd_delta(do, out, delta)
        _C.flash_bwd(*bwd_args, **bwd_kw)
    ms_f = _time_events(fwd, iters, warm=3)
    kinds["fwd"] = list(_C.last_launch_kinds())
    ms_b = _time_events(bwd, max(2, iters // 2), warm=2)
    kinds["bwd"] = list(_C.last_launch_kinds())
    F = fwd_flops(B, Hq, S, D)
    tf = lambda flops, ms: flops / (ms * 1e-3) / 1e12
    res = dict(fwd_ms=round(ms_f, 4), bwd_ms=round(ms_b, 4), fwd=tf(F, ms_f), bwd=tf(2.5 * F, ms_b),
               fwd_bwd=tf(3.5 * F, ms_f + ms_b), kinds=kinds)
    if per_kernel:
        n = max(2, iters // 2)
        res["delta_ms"] = round(_time_events(lambda: _C.bwd_delta(do, out, delta), n, warm=1), 4)
        res["dkdv_ms"] = round(_time_events(lambda: _C.flash_bwd(*bwd_args, only="dkdv", **bwd_kw), n, warm=1), 4)
        res["dq_ms"] = round(_time_events(lambda: _C.flash_bwd(*bwd_args, only="dq", **bwd_kw), n, warm=1), 4)
    if keep:
        res["tensors"] = dict(q=q, k=k, v=v, do=do, out=out, lse=lse, dq=dq, dk=dk, dv=dv)
    return res


def _layer_fwd_bwd(B, S, Hq, Hkv, D, dev, iters, warm=2, family=None):
    """One fwd+bwd step of the LAYER on one GPU: LongContextAttention.forward + out.backward through autograd (what the
    reference's own benchmark times, benchmark/benchmark_longctx.py:200-255), on the sequence-parallel group main() has
    set up (1 x 1 here).  Device events on torch's current stream; fresh N(0,1) leaves, gradients dropped per step.
    Returns ms per step -- autograd, final_grads, the delta launch and every host-side gap included.  `family`: the
    kernel family of every launch of the step (_C.set_kernel_family; the autograd thread reads the same default)."""
    import yunchang_amd as Y
    from yunchang_amd import _C
    q, k, v, do = _kernel_inputs(B, S, Hq, Hkv, D, dev, seed=2)
    for t in (q, k, v):
        t.requires_grad_(True)
    attn = Y.LongContextAttention(ring_impl_type="zigzag", attn_type=Y.AttnType.HIP)

    def step():
        attn(q, k, v, causal=True).backward(do)
        q.grad = k.grad = v.grad = None
    prev = _C.set_kernel_family(family or "auto")
    try:
        return _time_events(step, iters, warm=warm)
    finally:
        _C.set_kernel_family(prev)


def sampled_parity(t, n_rows=8, n_keys=8, seed=0):
    """Sampled rows / keys of a causal forward + backward (batch 0, first and last query head / KV head) against exact
    attention in fp64 on the device: out, LSE and dQ of `n_rows` query rows, dK and dV of `n_keys` keys (summed over the
    GQA group's query heads and over every row that sees the key).  Definitions as the block contract's
    (yunchang/kernels/attention.py:205-250): delta = rowsum(dO * out) with the 16-bit `out` the backward was given;
    the key columns take softmax probabilities from the kernel's fp32 LSE, which the row samples check against the
    exact one.  Size-independent: works at the metric's S = 65536.  Returns max-abs errors."""
    q, k, v, do, out, lse, dq, dk, dv = (t[n] for n in ("q", "k", "v", "do", "out", "lse", "dq", "dk", "dv"))
    B, S, Hq, D = q.shape
    Hkv = k.shape[2]
    G = Hq // Hkv
    scale = D ** -0.5
    rs = np.random.RandomState(seed)
    rows = sorted(r for r in {0, 255, 256, S // 2 - 1, S // 2, S - 1, *rs.randint(0, S, size=n_rows).tolist()} if 0 <= r < S)
    keys = sorted(j for j in {0, 127, 128, S - 1, *rs.randint(0, S, size=n_keys).tolist()} if 0 <= j < S)
    err = dict(out=0.0, lse=0.0, dq=0.0, dk=0.0, dv=0.0)
    # ... and the same errors in units of the parity tests' comparator (tests/golden_util.py: an element passes iff
    # |got - want| <= atol + rtol |want|): a ratio below 1 is a pass, NaN propagates
    tol = dict(out=(2e-2, 2e-2), lse=(2e-3, 1e-4), dq=(5e-2, 5e-2), dk=(5e-2, 5e-2), dv=(5e-2, 5e-2))
    ratio = dict(out=0.0, lse=0.0, dq=0.0, dk=0.0, dv=0.0)

    def up(name, a, b):
        d = (a.double() - b).abs()
        err[name] = nanmax(err[name], float(d.max()))
        ratio[name] = nanmax(ratio[name], float((d / (tol[name][0] + tol[name][1] * b.abs())).max()))
    for h in sorted({0, Hq - 1}):
        kd, vd = k[0, :, h // G].double(), v[0, :, h // G].double()
        for i in rows:
            sc = (kd[:i + 1] @ q[0, i, h].double()) * scale
            l_ref = torch.logsumexp(sc, 0)
            p = torch.exp(sc - l_ref)
            up("out", out[0, i, h], p @ vd[:i + 1])
            up("lse", lse[0, h, i], l_ref)
            doi = do[0, i, h].double()
            dp = vd[:i + 1] @ doi
            delta = (doi * out[0, i, h].double()).sum()
            up("dq", dq[0, i, h], ((p * (dp - delta)) @ kd[:i + 1]) * scale)
    for hk in sorted({0, Hkv - 1}):
        kd, vd = k[0, :, hk].double(), v[0, :, hk].double()
        for j in keys:
            rdk = torch.zeros(D, dtype=torch.float64, device=q.device)
            rdv = torch.zeros_like(rdk)
            for g in range(G):
                h = hk * G + g
                qi, doi = q[0, j:, h].double(), do[0, j:, h].double()
                p = torch.exp((qi @ kd[j]) * scale - lse[0, h, j:].double())
                rdv += p @ doi
                ds = p * (doi @ vd[j] - (doi * out[0, j:, h].double()).sum(-1))
                rdk += (ds @ qi) * scale
            up("dk", dk[0, j, hk], rdk)
            up("dv", dv[0, j, hk], rdv)
    return {"max_abs_err": {n: round(e, 6) for n, e in err.items()},
            "max_err_over_tolerance": {n: round(e, 4) for n, e in ratio.items()},
            "tolerance_atol_rtol": {n: list(v) for n, v in tol.items()}, "rows": len(rows), "keys": len(keys),
            "heads": sorted({0, Hq - 1}), "kv_heads": sorted({0, Hkv - 1}),
            "truth": "exact causal attention and its gradients in fp64 on the sampled rows / key columns"}


def mfma_ceiling(dev):
    """What the matrix pipe sustains on THIS box in THIS run (usp_mfma_probe: an MFMA-only loop of the flash kernels'
    instruction, no LDS / VALU / memory traffic) on N(0,1) bf16 operands -- the bench's data distribution -- and on zeros,
    at one and at two waves per SIMD; with the sustained shader clock of the loop (s_memtime / s_memrealtime ticks).
    The part clocks by power and MFMA power depends on the operand bits: the N(0,1) figure is the ceiling a flash kernel
    can approach here, the zeros figure shows how much of the nominal 2.5 PFLOP/s is a property of the data."""
    from yunchang_amd import _C
    g = torch.Generator(device=dev).manual_seed(3)
    ops = {"normal": torch.randn(1 << 20, device=dev, generator=g).to(torch.bfloat16),
           "zeros": torch.zeros(1 << 20, device=dev, dtype=torch.bfloat16)}
    clocks = torch.zeros(2, dtype=torch.int64, device=dev)
    res = {}
    for name, buf in ops.items():
        for w in (1, 2):
            iters = 2000 // w
            flops = [0.0]

            def launch():
                flops[0] = _C.mfma_probe(buf, iters, w, clocks)
            ms = _time_events(launch, 60, warm=60)              # ~0.15 s of heat, ~0.15 s timed
            c = clocks.tolist()
            res[f"{name}_w{w}"] = {"TFLOPs": round(flops[0] / (ms * 1e-3) / 1e12, 1),
                                   "clock_GHz": round(c[0] / max(1, c[1]) * 0.1, 3) if c[1] else None}
    best = max(res["normal_w1"]["TFLOPs"], res["normal_w2"]["TFLOPs"])
    return {"what": "MFMA-only loop (v_mfma_f32_32x32x16_bf16, 8 independent accumulators, 64 distinct operand pairs), one "
                    "workgroup per CU, w = waves per SIMD; TFLOP/s executed and the loop's sustained shader clock",
            **res, "sustained_ceiling_TFLOPs": best,
            "sustained_ceiling_note": "the better of the two N(0,1) figures: the rate a kernel made of nothing but MFMAs sustains "
                                      "on this box on the bench's data distribution"}


BENCH_ORDER_DEFAULT = "kernels_first"


def kernel_roofline(cfg, dev, traffic, iters=3):
    """The kernels of the N = 1 step (B1 S65536 H32/Hkv4 D128 causal fwd+bwd), each timed alone, live, with device events on
    the stream it is launched on; the layer-level step on both kernel families (same box, same run); the MFMA-only ceiling
    of this box; C2 (BASELINE configs[1], the headline of rounds 1-4) as a secondary block.  The roofline entry names the
    DOMINANT kernel of the step, flash_bwd_dkdv64_kernel.  `traffic`: pmc_traffic(), taken by the caller BEFORE any device
    work -- it may disassemble the library (seconds of host-only time), and nothing host-only may sit between this
    function's kernels and the timed region (the part would clock down again)."""
    B, S, Hq, Hkv, D = cfg["B"], cfg["S"], cfg["Hq"], cfg["Hkv"], cfg["D"]
    frac = lambda x: round(x / PEAK_BF16_TFLOPS, 4)
    tfs = lambda flops, ms: flops / (ms * 1e-3) / 1e12
    F = fwd_flops(B, Hq, S, D)
    _fwd_bwd_kernels(B, S, Hq, Hkv, D, dev, 2)                      # ~0.4 s of the kernels first: sustained clocks
    t = _fwd_bwd_kernels(B, S, Hq, Hkv, D, dev, iters, keep=True, per_kernel=True)
    try:                                                     # parity AT the metric's size (sampled: fp64 rows / key columns)
        parity = sampled_parity(t.pop("tensors"))
    except Exception as e:
        t.pop("tensors", None)
        parity = {"error": repr(e)[:200]}
    # the dK/dV launch computes S, dP, dV, dK: four of the backward's five algorithmic matmuls (all four are needed for dK and
    # dV whatever the formulation); the dQ launch is credited with its one new matmul (its S and dP are recompute: no credit)
    mm = 0.5 * F
    dkdv_tf, dq_tf = tfs(4 * mm, t["dkdv_ms"]), tfs(1 * mm, t["dq_ms"])
    ab = {}
    for fam in ("row64", "wave32"):                          # the layer step on both kernel families, same box, same run
        try:
            ab[fam] = round(_layer_fwd_bwd(B, S, Hq, Hkv, D, dev, 2, warm=1, family=fam), 3)
        except Exception as e:
            ab[fam] = None
            print(f"layer step on family {fam} failed to run: {e!r}", file=sys.stderr)
    try:
        ceiling = mfma_ceiling(dev)
    except Exception as e:
        ceiling = {"error": repr(e)[:200], "sustained_ceiling_TFLOPs": None}
    top = ceiling.get("sustained_ceiling_TFLOPs")
    of_ceiling = lambda executed_tf: None if not top else round(executed_tf / top, 4)
    roof = {"bound": "mfma",
            "kernel": "usp::flash_bwd_dkdv64_kernel<bf16,causal> (4 waves, one per SIMD: roles S/P/dV and dP/dS/dK on separate "
                      "SIMDs, 64 keys per wave) -- the dominant kernel of the step",
            "kernels_launched": t["kinds"],
            "achieved": round(dkdv_tf, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": frac(dkdv_tf),
            "kernel_ms": t["dkdv_ms"],
            "achieved_definition": "algorithmic FLOPs of one launch = 4 of the backward's 5 matmuls (S, dP, dV, dK: 2.0 x the "
                                   "forward's 4 B Hq S^2 D / 2) / the launch's duration (dK/dV kernel + its GQA head reduce)",
            "frac_of_sustained_ceiling": of_ceiling(dkdv_tf),
            "traffic": traffic,
            "step": {"shape_BSHD": [B, S, Hq, D], "kv_heads": Hkv, "pass": "fwd+bwd, causal",
                     "kernels": "flash_fwd64_kernel + delta_kernel + flash_bwd_dkdv64_kernel (+ reduce_heads_kernel) + flash_bwd_dq64_kernel",
                     "fwd_ms": t["fwd_ms"], "delta_ms": t["delta_ms"], "dkdv_ms": t["dkdv_ms"], "dq_ms": t["dq_ms"],
                     "bwd_ms": t["bwd_ms"], "iter_ms": round(t["fwd_ms"] + t["bwd_ms"], 3),
                     "fwd_achieved": round(t["fwd"], 1), "fwd_frac": frac(t["fwd"]),
                     "fwd_frac_of_sustained_ceiling": of_ceiling(t["fwd"]),
                     "dq_achieved": round(dq_tf, 1), "dq_executed": round(3 * dq_tf, 1),
                     "bwd_achieved": round(t["bwd"], 1), "bwd_frac": frac(t["bwd"]),
                     "bwd_executed": round(t["bwd"] * 7 / 5, 1), "bwd_executed_frac_of_sustained_ceiling": of_ceiling(t["bwd"] * 7 / 5),
                     "achieved": round(t["fwd_bwd"], 1), "frac": frac(t["fwd_bwd"]),
                     "note": "algorithmic FLOPs: backward = 2.5x forward (the two-launch backward executes 3.5x: S and dP are "
                             "computed in both launches)"},
            "layer_step_ms_by_kernel_family": {**ab, "note": "LongContextAttention fwd + backward through autograd, 2 timed steps "
                                               "each, every flash launch pinned to the family (ABI v6 USP_FORCE_*): row64 = one wave "
                                               "per SIMD (rounds 4-5), wave32 = two waves per SIMD (rounds 1-3)"},
            "mfma_ceiling": ceiling,
            "sampled_parity": parity}
    # secondary: BASELINE configs[1] (B2 S8192 H16 D128 causal), the headline of rounds 1-4
    try:
        c = C2
        _fwd_bwd_kernels(c["B"], c["S"], c["Hq"], c["Hkv"], c["D"], dev, 100)
        t2 = _fwd_bwd_kernels(c["B"], c["S"], c["Hq"], c["Hkv"], c["D"], dev, 20)
        t2w = _fwd_bwd_kernels(c["B"], c["S"], c["Hq"], c["Hkv"], c["D"], dev, 20, family="wave32")
        layer_ms = _layer_fwd_bwd(c["B"], c["S"], c["Hq"], c["Hkv"], c["D"], dev, 10)
        F2 = fwd_flops(c["B"], c["Hq"], c["S"], c["D"])
        roof["c2"] = {"workload": c["name"], "fwd_kernel_ms": t2["fwd_ms"], "fwd_achieved": round(t2["fwd"], 1), "fwd_frac": frac(t2["fwd"]),
                      "fwd_frac_of_sustained_ceiling": of_ceiling(t2["fwd"]),
                      "bwd_ms": t2["bwd_ms"], "bwd_achieved": round(t2["bwd"], 1), "bwd_frac": frac(t2["bwd"]),
                      "fwd_bwd_frac": frac(t2["fwd_bwd"]), "layer_ms": round(layer_ms, 4),
                      "layer_frac": frac(3.5 * F2 / (layer_ms * 1e-3) / 1e12),
                      "wave32_family": {"fwd_kernel_ms": t2w["fwd_ms"], "bwd_ms": t2w["bwd_ms"]},
                      "kernels_launched": t2["kinds"]}
    except Exception as e:                                   # informative entry: never kill the measurement
        roof["c2"] = {"error": repr(e)[:200]}
    return roof


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


def kernel_source_sha16():
    """Identity of the kernel sources a profile belongs to (tools/prof_round.sh writes it into the summary)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "long-context-attention_amd", "csrc")
    for name in ("usp_common.hpp", "usp_item_deal.h", "usp_mfma64.hpp", "usp_fwd_params.hpp", "usp_bwd_params.hpp",
                 "usp_flash_fwd.hip", "usp_flash_fwd64.hip", "usp_flash_bwd.hip", "usp_flash_bwd64.hip", "usp_flash_bwd_dq64.hip",
                 "Makefile"):
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


ROOFLINE_KERNEL = "flash_bwd_dkdv64_kernel"        # the dominant kernel of the N = 1 step
# algorithmic HBM bytes of ONE dK/dV launch at the N = 1 workload (B1 S65536 H32/Hkv4 D128 bf16): q and dO read once
# (2 x 512 MiB), k and v once (2 x 64 MiB), lse and delta (2 x 8 MiB fp32), dk and dv written once in 16 bits (2 x 64 MiB)
ROOFLINE_ALGORITHMIC_MB = round((2 * 65536 * 32 * 128 * 2 + 2 * 65536 * 4 * 128 * 2 + 2 * 32 * 65536 * 4 + 2 * 65536 * 4 * 128 * 2) / 1e6, 1)


def pmc_traffic():
    """HBM bytes per launch of the roofline kernel (flash_bwd_dkdv64_kernel at the N = 1 workload's shape) from the committed
    rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes; tools/prof_round.sh).  Hardware
    counters cannot be collected from inside this process, so the numbers come from profiles/r06_rocprof_summary.txt -- but
    ONLY if that profile was taken from the kernel sources of this tree (the summary carries their hash) or from the same
    MACHINE CODE of that kernel (tools/kernel_isa.py); otherwise null."""
    name = "r06_rocprof_summary.txt"
    path = os.path.join(ROOT, "profiles", name)
    try:
        rd = wr = sha = isa = None
        for ln in open(path):
            if ln.startswith("kernel_src_sha16:"):
                sha = ln.split(":")[1].strip()
            if ln.startswith("roofline_kernel_isa_sha16:"):
                isa = ln.split(":")[1].split()[0]
            if ROOFLINE_KERNEL in ln and "HBM read bytes/launch" in ln and rd is None:
                rd = float(ln.split("=")[1].split("MB")[0])
            if ROOFLINE_KERNEL in ln and "HBM write bytes/launch" in ln and wr is None:
                wr = float(ln.split("=")[1].split("MB")[0])
        if rd is None or wr is None:
            return None
        res = {"kernel": ROOFLINE_KERNEL, "read_MB": rd, "write_MB": wr, "algorithmic_MB": ROOFLINE_ALGORITHMIC_MB,
               "kernel_src_sha16": sha,
               "note": "above the algorithmic bytes by the flash tiling: every 128-key item re-streams its heads' Q / dO tiles.  The 32 workgroups an "
                       "XCD runs side by side share one stream through that XCD's L2, so the tiling's own floor is (items / 32) x the average "
                       "stream = 16384 / 32 x 16 MB = 8.2 GB per launch (the 1.36 GB would need ONE L2 for all 256 CUs); the rest is "
                       "workgroups drifting apart inside a pass, plus the fp32 dK / dV slabs of the two-heads-per-item launch that "
                       "reduce_heads_kernel sums.  3 % of the HBM roof over the launch: the kernel is MFMA-bound by 25x",
               "source": f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, same sources)"}
        if sha == kernel_source_sha16():
            return res
        # The sources have changed since the profile.  The figures still describe THIS library if the profiled kernel's
        # machine code is the same: tools/kernel_isa.py disassembles the shipped library and hashes that kernel's
        # instruction stream.
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import kernel_isa
            mine = kernel_isa.isa_identity(os.path.join(ROOT, "long-context-attention_amd", "libusp_hip.so"))[0]
        except Exception as e:                                   # no llvm-objdump, unreadable library ...
            mine = f"unavailable ({repr(e)[:80]})"
        if isa is not None and mine == isa:
            res["source"] = (f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE); the sources have "
                             "changed since, the machine code of the profiled kernel has not (tools/kernel_isa.py)")
            res["roofline_kernel_isa_sha16"] = isa
            res["kernel_src_sha16_now"] = kernel_source_sha16()
            return res
        # a stale number is worse than none: report where the last measurement is and what it was taken from
        return {"kernel": ROOFLINE_KERNEL, "read_MB": None, "write_MB": None, "algorithmic_MB": ROOFLINE_ALGORITHMIC_MB,
                "kernel_src_sha16": kernel_source_sha16(),
                "stale": f"profiles/{name} holds {rd} MB read + {wr} MB written per launch, taken from "
                         f"kernel sources {sha} (kernel machine code {isa}); this library's: {mine} -- no figure is claimed"}
    except OSError:
        return None


def cpu_baseline(cfg):
    """The reference's TORCH attention path on the host cores of this box, in the same run.  `value`: the
    restatement of what LongContextAttention(attn_type=TORCH_EFFICIENT) executes on one rank
    (oracle/ref_cpu_path.py: the layer's layout copies + the torch CPU attention op the path resolves to on a
    host), bf16, every core, on a BOUNDED sample of the N = 1 workload: the causal forward over the first 16384 tokens of
    the 65536 (causal attention over a prefix IS the workload's first rows; 1/16 of the forward's FLOPs), all 32 query
    heads, K/V heads expanded 4 -> 32 because that path has no GQA.  Forward only: the reference's TORCH
    backward raises "Not implemented" (yunchang/kernels/attention.py:138-159), so there is no CPU backward to time.
    Secondary: the plain C port of the algorithm (oracle/attn_oracle.c, fp32 in / fp64 accumulate, OpenMP over heads)."""
    from oracle import ref_cpu_path as R
    cores = os.cpu_count() or 1
    B, Hq, D = cfg["B"], cfg["Hq"], cfg["D"]
    S = min(cfg["S"], 16384)
    n = 2
    res = {"unit": "TFLOP/s", "cores": cores, "kind": "port", "value": None,
           "sample": f"causal forward over the first {S} of the workload's {cfg['S']} tokens (B={B} H={Hq} D={D}, bf16, K/V heads "
                     f"expanded to {Hq}: the path has no GQA), 1 warm-up + {n} timed passes on {cores} threads; "
                     f"oracle/ref_cpu_path.py = the reference's TORCH path on one rank "
                     f"(hybrid/attn_layer.py:111-158 -> ring_flash_attn.py:20-57 -> kernels/attention.py:44-136 with "
                     f"aten::_scaled_dot_product_flash_attention_for_cpu, the op that path needs on a host); forward only -- "
                     f"the reference's TORCH backward is not implemented (kernels/attention.py:138-159)"}
    try:
        torch.set_num_threads(cores)
        g = torch.Generator().manual_seed(0)
        q, k, v = (torch.randn((B, S, Hq, D), generator=g).to(torch.bfloat16) for _ in range(3))
        R.long_context_attention_forward_cpu(q, k, v, True)
        t0 = time.perf_counter()
        for _ in range(n):
            R.long_context_attention_forward_cpu(q, k, v, True)
        dt = (time.perf_counter() - t0) / n
        res["value"] = round(fwd_flops(B, Hq, S, D) / dt / 1e12, 5)
        res["seconds_per_pass"] = round(dt, 3)
    except Exception as e:                                   # pragma: no cover
        res["error"] = repr(e)[:200]
    # secondary: the C port, one (batch, head) problem per thread
    H, S = max(1, min(cores, Hq * B)), min(S, 4096)          # bounded: a few seconds
    rs = np.random.RandomState(0)
    q, k, v = (rs.standard_normal((1, S, H, D)).astype(np.float32) for _ in range(3))
    try:
        L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libattn_oracle.so"))
        out = np.empty_like(q)
        lse = np.empty((1, H, S), np.float32)
        fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        t0 = time.perf_counter()
        L.usp_oracle_attn_fwd(fp(q), fp(k), fp(v), 1, S, S, H, H, D, ctypes.c_float(D ** -0.5), 1, fp(out), fp(lse))
        dt = time.perf_counter() - t0
        res["c_port"] = {"value": round(fwd_flops(1, H, S, D) / dt / 1e12, 5), "threads": H, "seconds": round(dt, 2),
                         "what": f"oracle/attn_oracle.c, S={S}, fp32 in / fp64 accumulate, OpenMP over heads"}
    except OSError as e:
        res["c_port"] = {"value": None, "error": str(e)}
    return res


def exchange_mode(attn, lq, lk, cfg, ws):
    """How the layer moves q, k, v between the sequence and the head sharding (for the config record)."""
    if cfg["ud"] == 1:
        return "none (ulysses degree 1)"
    if not hasattr(attn, "_packed_exchange"):
        return "packed q|k|v, pipelined over head groups (AsyncLongContextAttention)"
    cap = attn._packed_exchange(lq, lk)
    if cap is None:
        return "three separate exchanges (reference structure)"
    from yunchang_amd.hybrid.async_attn_layer import _groups, _k_split_groups, _link_bound
    S = lq.shape[1] * cfg["ud"]
    lb = _link_bound(cfg["Hq"], cfg["Hkv"], cfg["ud"], cfg["B"], S, lq.shape[-1], lq.element_size(), cfg["rd"], True)
    ks = not cfg.get("bwd", False) and _k_split_groups(None, cfg["B"], S, True)
    ng = _groups(cfg["Hq"], cfg["Hkv"], cfg["ud"], cfg["B"], S, max_groups=cap, link_bound=lb, k_split=ks)[0]
    return (f"one packed q|k|v exchange per head group, {ng} group(s)" + (", pipelined on a side stream" if ng > 1 else "")
            + (", sized for the forward K split" if ks else ""))


def _never_fatal(fn, *a):
    """A descriptive field must not cost the measurement its line."""
    try:
        return fn(*a)
    except Exception as e:
        return {"error": repr(e)[:200]}


def run_env():
    """Every environment switch that can change what this run measures (the library reads USP_*; RCCL / HSA / HIP read theirs)."""
    keys = sorted(k for k in os.environ if k.startswith(("USP_", "NCCL_", "RCCL_", "HSA_", "HIP_", "GPU_MAX_HW_QUEUES", "TORCH_NCCL_")))
    return {k: os.environ[k] for k in keys}


def derived_decisions(cfg, attn, lq, lk, ws):
    """What the library DERIVED for this configuration, with the figures it derived it from and where each figure came from
    (a constant, an environment pin, or a probe at set_seq_parallel_pg): so that a scaling line can be read without the source."""
    import yunchang_amd.hybrid.async_attn_layer as AL
    from yunchang_amd.comm import link as L
    import yunchang_amd.comm.relay_exchange as RX
    ud, rd = cfg["ud"], cfg["rd"]
    S = lq.shape[1] * ud
    d = {
        "link_rate_GBs": round(L.link_bytes_per_s() / 1e9, 1),
        "link_rate_source": ("USP_LINK_GBS" if os.environ.get("USP_LINK_GBS") else
                             "measured at set_seq_parallel_pg (comm/link.py, MIN over ranks)" if L.measured() else
                             "constant (no probe: USP_LINK_PROBE != 1, one rank, or not RCCL)"),
        "kernel_rate_TFs": round(L.kernel_flops_per_s() / 1e12, 1),
        "kernel_rate_source": ("USP_KERNEL_TFS" if os.environ.get("USP_KERNEL_TFS") else
                               "measured at set_seq_parallel_pg (forward kernel, B1 S4096 H16, MIN over ranks)" if L.kernel_measured()
                               else "constant (no probe)"),
        "device_cus": L.device_cus(),
        "head_group_fill_items": {"default": AL.fill_items(), "link_bound": AL.fill_items(True), "link_bound_k_split": AL.fill_items(True, True),
                                  "source": "2 / 1 / 0.5 work items per CU x device_cus" if AL._FILL_ITEMS is None else "pinned"},
        "safe_comm": bool(AL.safe_comm()),
        "pipeline_beside_ring": bool(AL.pipeline_mode(rd)) if ud > 1 else None,
        "exchange_relay": bool(RX.relay_enabled()),
    }
    if ud > 1:
        lb = AL._link_bound(cfg["Hq"], cfg["Hkv"], ud, cfg["B"], S, lq.shape[-1], lq.element_size(), rd, True)
        ks = not cfg.get("bwd", False) and AL._k_split_groups(None, cfg["B"], S, True)
        d["exchange_link_bound"] = bool(lb)
        d["forward_k_split_groups"] = bool(ks)
        d["head_groups"] = AL._groups(cfg["Hq"], cfg["Hkv"], ud, cfg["B"], S, link_bound=lb, k_split=ks)[0]
        pipelined = bool(AL.pipeline_mode(rd))
        d["self_chunk_start"] = bool(AL.self_chunk_mode(ud, rd, True, cfg["impl"], lq.shape[1])) and (pipelined or rd == 1)
        d["self_chunk_all_groups"] = bool(AL.self_chunk_all_groups())
        _, kvh_g, g_g = AL._groups(cfg["Hq"], cfg["Hkv"], ud, cfg["B"], S, link_bound=lb, k_split=ks)
        d["tail_row_pieces"] = int(AL.tails_mode(ud, rd, True, cfg["impl"], lq.shape[1], pipelined,
                                                 cfg["B"] * kvh_g * g_g * ((lq.shape[1] + 255) // 256)))    # (> 0: also dq ahead of dk | dv)
    return d


def barrier(ws):
    if ws > 1:            # a barrier over one rank is empty; NCCL would still launch an all-reduce for it
        if dist.get_backend() == "nccl":      # name the device: without it ProcessGroupNCCL guesses one from the rank
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def _sync(dev):
    if dev.type == "cuda":
        torch.cuda.synchronize()


def timed(step, steps, ws, dev, device_ms=None, spread=None):
    """K steps bracketed by barrier + synchronize on both sides; max over ranks; seconds per step (host clock).
    `device_ms`: a list that receives this rank's per-step time between two device events recorded on the compute
    stream around the same K steps (diagnostic: host clock minus device clock = launch latency of the first step +
    the wake-up of the final synchronize).  `spread`: a dict that receives the fastest and the slowest rank's OWN time
    for the K steps (its clock stopped at its own synchronize, in front of the closing barrier), ms per step."""
    _sync(dev)
    barrier(ws)
    _sync(dev)
    ev = None
    if device_ms is not None and dev.type == "cuda":
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    t0 = time.perf_counter()
    if ev:
        ev[0].record()
    for _ in range(steps):
        step()
    if ev:
        ev[1].record()
    _sync(dev)
    own = time.perf_counter() - t0
    barrier(ws)
    _sync(dev)
    dt = time.perf_counter() - t0
    if ev:
        device_ms.append(ev[0].elapsed_time(ev[1]) / steps)
    lo = hi = own
    if ws > 1:
        tmax = torch.tensor([dt, own, -own], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt, hi, lo = float(tmax[0].item()), float(tmax[1].item()), -float(tmax[2].item())
    if spread is not None:
        spread.update(min=round(lo / steps * 1e3, 4), max=round(hi / steps * 1e3, 4))
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
        # every receive becomes a local copy of the same size from one of this rank's own send buffers (the batches of
        # the mesh fetches are not (send, recv) pairs, and ranks post different numbers of each); sends are dropped
        sends = [op.tensor for op in self._ops if op.op is dist.isend]
        for i, op in enumerate(o for o in self._ops if o.op is dist.irecv):
            src = next((t for t in sends[i % max(1, len(sends)):] + sends if t.shape == op.tensor.shape), None)
            if src is not None:
                op.tensor.copy_(src)
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


class _LineOnce:
    """Prints rank 0's JSON line exactly once, from whichever thread gets there first (None on other ranks)."""

    def __init__(self, line):
        self.line, self._lock, self.done = line, threading.Lock(), False

    def __call__(self, extra=None):
        with self._lock:
            if self.done:
                return
            self.done = True
            if self.line is not None:
                print(json.dumps({**self.line, **(extra or {})}), flush=True)


class _Deadline:
    """Context manager: if the body has not finished after `seconds`, call emit(extra) and leave the process with exit
    code 0 (every rank arms the same deadline, so a rank stuck in a collective cannot keep the launcher waiting for
    the process-group timeout)."""

    def __init__(self, seconds, emit, extra):
        self._t = threading.Timer(seconds, self._fire)
        self._t.daemon = True
        self._emit, self._extra = emit, extra

    def _fire(self):
        self._emit(self._extra)
        sys.stdout.flush()
        os._exit(0)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._t.cancel()
        return False


def main(argv=None, dev=None):
    """`argv` / `dev`: the CPU tests drive this function in-process on host tensors (gloo group already set up, a
    test block backend installed); the driver calls it with neither."""
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
    args = ap.parse_args(argv)

    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if ws != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ws}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if args.gpus not in WORKLOADS:
        raise SystemExit(f"--gpus must be one of {sorted(WORKLOADS)}")
    per_config = os.environ.get("USP_BENCH_WORKLOAD", "metric") == "configs"      # rounds 1-4: BASELINE's config per GPU count
    cfg = dict((CONFIG_WORKLOADS if per_config else WORKLOADS)[args.gpus])
    if args.bwd >= 0:
        cfg["bwd"] = bool(args.bwd)
    # USP_BENCH_BACKEND=gloo is a DEVELOPMENT smoke mode, not a measurement: all ranks share cuda:0 and
    # talk over gloo, so the N > 1 code path of this script can be exercised on a 1-GPU box (RCCL refuses
    # two ranks on one device).  The line it prints is tagged "smoke" and must not be read as a result.
    backend = os.environ.get("USP_BENCH_BACKEND", "nccl")
    smoke = backend != "nccl"
    if smoke:
        local_rank = 0
    if dev is None:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if ws == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29751")
    own_pg = not dist.is_initialized()
    if own_pg:
        dist.init_process_group(backend, rank=rank, world_size=ws)

    if ws > 1:                            # the benchmark harness opts in to the link-rate probe (comm/link.py): every rank
        os.environ.setdefault("USP_LINK_PROBE", "1")    # has set its device above, and the line records what was measured
    import yunchang_amd as Y
    if smoke and dev.type == "cuda":      # gloo's p2p is not stream-ordered for device tensors (tests/test_gpu_multiproc.py)
        import yunchang_amd.ring.utils as _U
        _commit = _U.RingComm.commit
        _U.RingComm.commit = lambda self: (torch.cuda.synchronize(), _commit(self))[1]
    Y.set_seq_parallel_pg(cfg["ud"], cfg["rd"], rank, ws)
    # two communicators (ulysses x ring grid): everything up to and including the first measurement runs in the SAFE
    # mode (one communicator in flight at a time); the overlapped mode is tried afterwards, under a deadline (below)
    import yunchang_amd.hybrid.async_attn_layer as AL
    two_comms = cfg["ud"] > 1 and cfg["rd"] > 1 and not args.async_ulysses and "USP_PIPELINE_ULYSSES" not in os.environ \
        and "USP_SAFE_COMM" not in os.environ
    if two_comms:
        AL._COMM_OVERRIDE.update(safe=True)
    # ulysses degree 2 at ring degree 1 (the 2-GPU grid): ONE communicator, but round 6's defaults (self-chunk start, row-chunked
    # tails, dq ahead of dk | dv) have never met two devices either -- the same staging: round 5's schedule first, the default
    # under a deadline behind it
    plain_first = cfg["ud"] == 2 and cfg["rd"] == 1 and not args.async_ulysses and "USP_SELF_CHUNK" not in os.environ \
        and "USP_TAILS" not in os.environ and "USP_PIPELINE_ULYSSES" not in os.environ
    if plain_first:
        AL._COMM_OVERRIDE.update(self_chunk="0", tails="0")
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

    parity_op = parity_rows = None
    out = step()
    if not args.no_parity:
        try:
            parity_op, parity_rows = parity_check(cfg, rank, ws, out.detach(), q, k, v)
        except Exception as e:                      # never let the optional check kill the measurement
            print(f"[rank {rank}] parity check failed to run: {e!r}", file=sys.stderr)
            parity_rows = float("nan")
        if ws > 1:
            pt = torch.tensor([float("nan") if parity_op is None else parity_op, parity_rows], device=dev)
            dist.all_reduce(pt, op=dist.ReduceOp.MAX)
            parity_op, parity_rows = (None if parity_op is None else float(pt[0].item())), float(pt[1].item())
    del q, k, v, do, out

    # Order of what follows (round 3; the driver's W=5 / K=20 run is 12 ms of device work): everything that is HOST-ONLY
    # comes first -- the PMC-traffic lookup (it may disassemble the library: seconds) and the collector freeze (a full
    # collection walks ~170 k torch objects: 40-60 ms, tools/host_cost.py) -- then device work only, back to back:
    # the kernel timings (N = 1, ~1 s), ~0.3 s of the step itself, the W warm-up steps, the K timed steps.  Round 2 had
    # the lookup and the freeze BEHIND the kernel timings: the part sat idle for seconds, clocked down, and the driver's
    # 20 steps read 22 % slower than the kernel they consist of (BENCH_r02.json: 0.5813 vs 0.4768 ms).
    traffic = pmc_traffic() if (ws == 1 and rank == 0) else None
    gc.collect()
    gc.freeze()          # the documented pattern: the collector keeps running, over new objects only

    # USP_BENCH_ORDER (N = 1): "kernels_first" = rounds 3-5, the kernel timings of the `roofline` block in front of the timed steps;
    # "headline_first" = the W warm-up + K timed steps FIRST, the diagnostics (which include the MFMA-only ceiling probe: the
    # hottest loop the part can run, and the slower kernel family's steps) behind them.
    order = os.environ.get("USP_BENCH_ORDER", BENCH_ORDER_DEFAULT)
    roofline = None
    if ws == 1 and rank == 0 and order != "headline_first":
        roofline = kernel_roofline(cfg, dev, traffic)
    flops = fwd_flops(cfg["B"], cfg["Hq"], cfg["S"], cfg["D"]) * (3.5 if cfg["bwd"] else 1.0)

    rank_spread = {}                        # the fastest / slowest rank's own ms per step of the LAST measure() call

    def measure():
        """W warm-up steps, K timed steps, back to back -> (ms per step, device-event ms).  Nothing else: the driver's
        --warmup describes the conditions (rounds 3-4 put up to 400 untimed steps in front; a 64K step is 15-120 ms, five of
        them are past any clock ramp, and at N = 1 the kernel timings run right in front)."""
        for _ in range(args.warmup):
            step()
        dms = []
        sec = timed(step, args.steps, ws, dev, dms, rank_spread)
        return sec * 1e3, (round(dms[0], 4) if dms else None)

    from yunchang_amd.comm import link as _link

    def make_line(ms, dms, comm_mode):
        value = flops / (ms * 1e-3) / 1e12
        if rank != 0:
            return None
        line = {
            "metric": "attention TFLOP/s (algorithmic, causal fwd+bwd) + iter ms of LongContextAttention at seqlen 64K, ulysses x ring"
                      if not per_config else "attention TFLOP/s (algorithmic, causal) of LongContextAttention ulysses x ring",
            "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": ws, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
            "scaling": "weak" if per_config else "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": cfg["name"], "global_shape_BSHD": [cfg["B"], cfg["S"], cfg["Hq"], cfg["D"]],
                       "kv_heads": cfg["Hkv"], "parallelism": f"ulysses{cfg['ud']}xring{cfg['rd']}",
                       "layout": cfg["impl"], "pass": "fwd+bwd" if cfg["bwd"] else "fwd",
                       "layer": "AsyncLongContextAttention" if args.async_ulysses else "LongContextAttention",
                       "ulysses_exchange": exchange_mode(attn, lq, lk, cfg, ws),
                       "comm_mode": comm_mode,
                       "env": run_env(),
                       "derived": _never_fatal(derived_decisions, cfg, attn, lq, lk, ws),
                       "tokens_per_gpu": cfg["S"] * cfg["B"] // ws,
                       "assumed": "B=1 and causal=True where BASELINE.json's config string is silent",
                       "iter_ms": round(ms, 4),
                       "host": "host-only work (gc.collect + gc.freeze, PMC lookup) first, then device work only: kernel "
                               "timings (N = 1), W warm-up steps, K timed steps"},
            "ms_per_step_device_events": dms,
            "ms_per_step_rank_min_max": dict(rank_spread),
            "frac_of_mfma_roofline": round(value / (ws * PEAK_BF16_TFLOPS), 4),
            "parity_max_abs_err_vs_reference_op": parity_op,
            "parity_max_abs_err_vs_fp64_rows": parity_rows,
        }
        if smoke:
            line["smoke"] = f"backend={backend}, all ranks on {dev} -- NOT a measurement"
        return line

    # A ulysses x ring grid has TWO communicators.  USP_SAFE_COMM=1 keeps one of them in flight at a time beside a ring
    # (one packed exchange in front of the ring attention, one behind it); the library's default pipelines the exchange
    # over head groups beside the ring traffic (both communicators in flight, each on its own side stream).  The
    # default has run through RCCL on one device only (tests/test_gpu_rccl_order.py), so the measurement is staged: the safe mode is measured FIRST
    # and its line is complete; the overlapped mode then runs under a deadline -- if it finishes, the faster of the two
    # is reported (and named, with the other's figure beside it); if it stalls, rank 0 prints the safe line and every
    # rank leaves with exit code 0.
    ms, dms = measure()
    comm_mode = ("safe: one communicator in flight at a time" if (two_comms or AL.safe_comm()) else
                 "plain: head-group pipeline without the self-chunk start and the tails (round 5's schedule)" if plain_first else
                 "library default" + (" (USP_PIPELINE_ULYSSES=%s)" % os.environ["USP_PIPELINE_ULYSSES"]
                                      if "USP_PIPELINE_ULYSSES" in os.environ else ""))
    line = make_line(ms, dms, comm_mode)
    if two_comms:
        safe_ms = ms
        fallback = _LineOnce(None if line is None else
                             {**line, "comm_mode_note": "the overlapped mode (the library default) did not finish "
                                                        "before its deadline; this is the safe mode's measurement"})
        budget = float(os.environ.get("USP_BENCH_MODE_DEADLINE_S", str(60 + 20 * safe_ms * 1e-3 * (args.warmup + args.steps))))
        with _Deadline(budget, fallback, None):
            AL._COMM_OVERRIDE.update(safe=False, pipeline="1")
            ms2, dms2 = measure()
        if ms2 < ms:
            ms, dms = ms2, dms2
            line = make_line(ms, dms, "overlapped (the library default): Ulysses exchange pipelined over head groups beside "
                                      "the ring, two communicators in flight; head groups start on the rank's own rows, the last "
                                      "group's output leaves in row pieces (USP_SELF_CHUNK / USP_TAILS defaults)")
        else:                                       # the overlap probe below runs in the mode that was reported
            AL._COMM_OVERRIDE.update(safe=True)
            AL._COMM_OVERRIDE.pop("pipeline", None)
        modes = {"safe": round(safe_ms, 4), "overlapped": round(ms2, 4)}
        # Third, on top of the faster of the two: the pair exchanges of the ulysses-2 grid striped over the idle links of
        # the mesh (comm/relay_exchange.py: two grouped send/recv phases on the world group) -- same deadline guard.
        import yunchang_amd.comm.relay_exchange as RX
        if cfg["ud"] == 2 and "USP_EXCHANGE_RELAY" not in os.environ:
            fallback = _LineOnce(None if line is None else
                                 {**line, "comm_modes_ms_per_step": modes,
                                  "comm_mode_note": "the relayed pair exchange (USP_EXCHANGE_RELAY=1) did not finish before its "
                                                    "deadline; this is the measurement without it"})
            with _Deadline(budget, fallback, None):
                RX._OVERRIDE["relay"] = True
                ms3, dms3 = measure()
            modes["relayed"] = round(ms3, 4)
            if ms3 < ms:
                ms, dms = ms3, dms3
                line = make_line(ms, dms, (line["config"]["comm_mode"] if line else "") +
                                 " + pair exchanges striped over the mesh (USP_EXCHANGE_RELAY=1)")
            else:
                RX._OVERRIDE.clear()
        if line is not None:
            line["comm_modes_ms_per_step"] = modes
    if plain_first:
        modes = {"plain": round(ms, 4)}
        fallback = _LineOnce(None if line is None else
                             {**line, "comm_modes_ms_per_step": modes,
                              "comm_mode_note": "the library default (self-chunk start + row-chunked tails) did not finish before its "
                                                "deadline; this is the measurement of round 5's schedule"})
        budget = float(os.environ.get("USP_BENCH_MODE_DEADLINE_S", str(60 + 20 * ms * 1e-3 * (args.warmup + args.steps))))
        with _Deadline(budget, fallback, None):
            AL._COMM_OVERRIDE.pop("self_chunk", None)
            AL._COMM_OVERRIDE.pop("tails", None)
            ms2, dms2 = measure()
        modes["default"] = round(ms2, 4)
        if ms2 < ms:
            ms, dms = ms2, dms2
            line = make_line(ms, dms, "library default: head groups start on the rank's own rows, the last group's output leaves in row "
                                      "pieces, its dq ahead of dk | dv (USP_SELF_CHUNK / USP_TAILS defaults)")
        else:
            AL._COMM_OVERRIDE.update(self_chunk="0", tails="0")
        if line is not None:
            line["comm_modes_ms_per_step"] = modes
    # (Round 5 measured the self-chunk start as one more deadline-guarded mode.  Since round 6 it is part of the library default
    # -- with the row-chunked tails beside a zigzag ring -- so the staged modes are: safe -> default -> + relayed pair exchange;
    # USP_SELF_CHUNK=0 / USP_TAILS=0 in the environment restore round 5's schedule for an A/B run.)
    value = flops / (ms * 1e-3) / 1e12

    # The measurement is complete here.  What follows is informative and must never cost the line: the overlap probe
    # re-runs the step with patched transports on every rank, so it runs under a deadline -- if it has not returned
    # by then (a rank stuck in a collective), rank 0 prints the line without it and every rank leaves.
    emit = _LineOnce(line)
    if ws > 1 and not args.no_overlap:
        with _Deadline(float(os.environ.get("USP_BENCH_PROBE_DEADLINE_S", "120")), emit,
                       {"overlap": {"value": None, "error": "overlap probe did not finish before its deadline"}}):
            try:
                overlap = overlap_probe(step, max(3, args.steps // 4), ws, dev, ms * 1e-3)
            except Exception as e:
                print(f"[rank {rank}] overlap probe failed to run: {e!r}", file=sys.stderr)
                overlap = {"value": None, "error": repr(e)}
        if rank == 0:
            line["overlap"] = overlap

    if rank == 0 and ws == 1:
        if roofline is None:
            roofline = kernel_roofline(cfg, dev, traffic)
        line["roofline"] = roofline
        line["config"]["order"] = order
        line["reference_kernel_on_this_gpu"] = reference_kernel(cfg, dev, (roofline.get("step") or {}).get("fwd_achieved", roofline["achieved"]))
        if cfg["bwd"]:
            line["reference_fwdbwd_on_this_gpu"] = reference_fwdbwd(cfg, dev, value)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg)
    emit()
    barrier(ws)
    AL._COMM_OVERRIDE.clear()            # (the staged measurement set the communicator mode in-process)
    import yunchang_amd.comm.relay_exchange as _RX
    _RX._OVERRIDE.clear()
    if own_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
