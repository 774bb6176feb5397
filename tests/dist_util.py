"""Spawn helpers for the multi-process CPU (gloo) tests."""
import os
import socket
import sys
import traceback

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _entry(rank, ws, port, fn, args, ret):
    try:
        for p in (ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.set_num_threads(1)
        dist.init_process_group("gloo", rank=rank, world_size=ws)
        ret[rank] = ("ok", fn(rank, ws, *args))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        ret[rank] = ("err", traceback.format_exc())
        raise


def run_distributed(fn, ws, *args):
    """Run fn(rank, ws, *args) on `ws` gloo ranks; returns the list of per-rank return values."""
    mgr = mp.Manager()
    ret = mgr.dict()
    try:
        mp.spawn(_entry, args=(ws, _free_port(), fn, args, ret), nprocs=ws, join=True)
    except Exception as e:
        msgs = [f"rank {r}: {v[1]}" for r, v in sorted(ret.items()) if v[0] == "err"]
        raise AssertionError("distributed run failed:\n" + "\n".join(msgs or [str(e)]))
    return [ret[r][1] for r in range(ws)]
