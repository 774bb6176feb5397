"""The xGMI link rate the schedules of this package size themselves by (head groups of the pipelined Ulysses exchange,
hybrid/async_attn_layer.py:_link_bound).

Rounds 1-2 used a constant (64 GB/s per link and direction, arithmetic on the MI355X guide's 7 links x ~153 GB/s
bidirectional).  `probe_link_rate` MEASURES it once per process, at `set_seq_parallel_pg` time -- a collective moment by
contract (every rank calls it, globals.py:22-81): a few 16 MiB send/recv rounds around the ring of ranks, timed with device
events, reduced with MIN over all ranks so that every rank sizes its head groups from the SAME number (ranks that
disagreed about a group count would post different collectives).  USP_LINK_GBS=<GB/s> pins the figure and skips the probe;
USP_LINK_PROBE=0 keeps the constant."""
import os

import torch
import torch.distributed as dist

DEFAULT_BYTES_PER_S = 64e9
_measured = None          # bytes/s, one link, one direction


def link_bytes_per_s() -> float:
    env = os.environ.get("USP_LINK_GBS")
    if env:
        try:
            return float(env) * 1e9
        except ValueError:
            pass
    return _measured if _measured else DEFAULT_BYTES_PER_S


def measured() -> bool:
    return _measured is not None


def probe_link_rate(rank: int, world_size: int, nbytes: int = 16 << 20, rounds: int = 4):
    """Collective over the default group.  Only on RCCL with more than one rank and a GPU; anywhere else (gloo tests, one
    rank) the constant stays.  Returns the rate in bytes/s or None."""
    global _measured
    if os.environ.get("USP_LINK_GBS") or os.environ.get("USP_LINK_PROBE", "1") == "0":
        return None
    if world_size < 2 or not dist.is_initialized() or dist.get_backend() != "nccl" or not torch.cuda.is_available():
        return None
    dev = torch.device("cuda", torch.cuda.current_device())
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    to, frm = (rank + 1) % world_size, (rank - 1) % world_size

    def hop():
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, src, to), dist.P2POp(dist.irecv, dst, frm)]):
            req.wait()
    for _ in range(2):
        hop()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(rounds):
        hop()
    e1.record()
    e1.synchronize()
    rate = torch.tensor([rounds * nbytes / max(e0.elapsed_time(e1) * 1e-3, 1e-9)], dtype=torch.float64, device=dev)
    dist.all_reduce(rate, op=dist.ReduceOp.MIN)         # one figure for every rank
    _measured = float(rate.item())
    return _measured
