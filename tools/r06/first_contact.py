"""First contact with a multi-GPU box, stage by stage (launched by tools/r06/first_contact.sh under torchrun, one process per
GPU; every stage under the shell's own deadline, every stage writes ONE JSON file on rank 0).

    --stage link     RCCL over xGMI: neighbour send/recv rate around the ring of ranks (comm/link.py's probe, MIN over ranks) and
                     a pair all_to_all_single at the sizes the 8-GPU grid exchanges; per-rank shader clocks are not read (no
                     telemetry the boxes report is trustworthy: DESIGN.md 4.6)
    --stage parity   the real layer on the real grid at a small size (S = 2048 per rank), forward + backward, in the mode given
                     by --mode (safe | default | relay): every rank's shard against exact fp64 attention computed on rank 0's host
                     and broadcast -- the first time any byte of this package crosses between two devices
    --backend gloo   DEVELOPMENT: the same stages on CPU tensors with the test oracle as block kernel (tests/test_bench_helpers.py
                     runs them so); not a measurement

The bench stages (safe, then the staged default / relay modes) are bench.py itself: see first_contact.sh."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch
import torch.distributed as dist

GRIDS = {1: (1, 1), 2: (2, 1), 4: (1, 4), 8: (2, 4)}          # (ulysses, ring) per GPU count: BASELINE's grids (bench.py WORKLOADS)


def stage_link(rank, ws, dev, cuda):
    res = {"stage": "link", "world_size": ws}
    if cuda:
        os.environ["USP_LINK_PROBE"] = "1"
        from yunchang_amd.comm import link
        rate = link.probe_link_rate(rank, ws)
        res["neighbour_send_recv_GBs"] = None if not rate else round(rate / 1e9, 2)
        res["kernel_rate_TFs"] = round(link.kernel_flops_per_s() / 1e12, 1)
    if ws >= 2:            # the pair exchange of ulysses degree 2: ranks (2i, 2i + 1), 20 MiB per direction (one head group of the 8-GPU grid)
        pairs = [dist.new_group([i, i + 1]) for i in range(0, ws - ws % 2, 2)]
        if rank < ws - ws % 2:
            grp = pairs[rank // 2]
            n = (20 << 20) if cuda else (1 << 16)
            send = torch.zeros(2, n, dtype=torch.uint8, device=dev)
            recv = torch.empty_like(send)
            for _ in range(2):
                dist.all_to_all_single(recv, send, group=grp)
            if cuda:
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                dist.all_to_all_single(recv, send, group=grp)
            if cuda:
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            mine = torch.tensor([n / dt / 1e9], dtype=torch.float64, device=dev)
        else:
            mine = torch.tensor([float("inf")], dtype=torch.float64, device=dev)
        dist.all_reduce(mine, op=dist.ReduceOp.MIN)
        res["pair_all_to_all_GBs_per_direction_min_over_pairs"] = round(float(mine.item()), 2)
    return res


def stage_parity(rank, ws, dev, cuda, mode):
    import yunchang_amd as Y
    import yunchang_amd.hybrid.async_attn_layer as AL
    import yunchang_amd.comm.relay_exchange as RX
    from oracle import usp_oracle as O
    ud, rd = GRIDS[ws]
    if not cuda:
        from yunchang_amd.kernels import set_block_backend
        from oracle_backend import OracleBlockBackend
        set_block_backend(OracleBlockBackend())
        AL._FILL_ITEMS = 1
    AL._COMM_OVERRIDE.clear()
    RX._OVERRIDE.clear()
    if mode == "safe":
        AL._COMM_OVERRIDE.update(safe=True)
    elif mode == "relay":
        RX._OVERRIDE["relay"] = True
    Y.set_seq_parallel_pg(ud, rd, rank, ws)
    B, Hq, Hkv, D = 1, 32 if cuda else 8, 4, 128 if cuda else 32
    S = (2048 if cuda else 32) * ws
    impl = "zigzag" if rd > 1 else "basic"
    torch.manual_seed(0)
    q, k, v, do = (torch.randn(B, S, h, D).to(torch.bfloat16) for h in (Hq, Hkv, Hkv, Hq))     # same seed on every rank
    ext = Y.EXTRACT_FUNC_DICT[impl]
    lq, lk, lv, ldo = (ext(t, rank, world_size=ws, rd=rd, ud=ud).detach().clone().to(dev) for t in (q, k, v, do))
    for t in (lq, lk, lv):
        t.requires_grad_(True)
    attn = Y.LongContextAttention(ring_impl_type=impl, attn_type=Y.AttnType.HIP)
    t0 = time.perf_counter()
    out = attn(lq, lk, lv, causal=True)
    out.backward(ldo)
    if cuda:
        torch.cuda.synchronize()
    sec = time.perf_counter() - t0
    # truth: exact attention and its gradients in fp64 (numpy on the host; S <= 16384), this rank's shard of it
    qn, kn, vn, don = (t.float().numpy().astype(np.float64) for t in (q, k, v, do))
    ro, rl = O.attention_ref(qn, kn, vn, causal=True)
    truth = (ro,) + tuple(O.block_bwd(don, qn, kn, vn, ro, rl, None, True))
    errs = []
    for got, want, tol in zip((out, lq.grad, lk.grad, lv.grad), truth, (2e-2, 5e-2, 5e-2, 5e-2)):
        w = ext(torch.from_numpy(np.ascontiguousarray(want)), rank, world_size=ws, rd=rd, ud=ud).float()
        d = (got.detach().float().cpu() - w).abs()
        errs.append(float((d / (tol + tol * w.abs())).max()))
    worst = torch.tensor(errs, dtype=torch.float64, device=dev)
    dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    worst = [round(float(x), 4) for x in worst.tolist()]
    AL._COMM_OVERRIDE.clear()
    RX._OVERRIDE.clear()
    return {"stage": "parity", "mode": mode, "grid": f"ulysses{ud}xring{rd}", "shape_BSHD": [B, S, Hq, D], "kv_heads": Hkv,
            "max_err_over_tolerance": dict(zip(("out", "dq", "dk", "dv"), worst)), "ok": max(worst) < 1.0,
            "first_step_s": round(sec, 3)}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", choices=["link", "parity"], required=True)
    ap.add_argument("--mode", default="safe", choices=["safe", "default", "relay"])
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--out", default=None)
    a = ap.parse_args(argv)
    rank, ws = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    cuda = a.backend == "nccl"
    if cuda:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dev = torch.device("cuda", torch.cuda.current_device())
    else:
        dev = torch.device("cpu")
    own = not dist.is_initialized()
    if own:
        dist.init_process_group(a.backend, rank=rank, world_size=ws)
    res = stage_link(rank, ws, dev, cuda) if a.stage == "link" else stage_parity(rank, ws, dev, cuda, a.mode)
    res["backend"] = a.backend
    if rank == 0:
        print(json.dumps(res), flush=True)
        if a.out:
            with open(a.out, "w") as f:
                json.dump(res, f)
    if own:
        dist.barrier()
        dist.destroy_process_group()
    return res


if __name__ == "__main__":
    main()
