"""The xGMI link rate the schedules of this package size themselves by (head groups of the pipelined Ulysses exchange,
hybrid/async_attn_layer.py:_link_bound).

The default is a constant (64 GB/s per link and direction, arithmetic on the MI355X guide's 7 links x ~153 GB/s
bidirectional).  `probe_link_rate` can MEASURE it once per process, at `set_seq_parallel_pg` time -- a collective moment by
contract (every rank calls it, globals.py:22-81): a few 16 MiB send/recv rounds around the ring of ranks, timed with device
events, reduced with MIN over all ranks so that every rank sizes its head groups from the SAME number (ranks that
disagreed about a group count would post different collectives).

The probe is OPT-IN (USP_LINK_PROBE=1; bench.py opts in for its multi-GPU runs): a hidden collective inside
`set_seq_parallel_pg` surprises scripts written for the reference -- its own tests never call torch.cuda.set_device, so
every rank's probe buffer would land on cuda:0 and RCCL would fail or hang at set-up -- it creates point-to-point
communicators nobody asked for, and a measured figure makes the head-group schedule vary from run to run.  When it runs,
the ranks first AGREE to run it (one MIN all-reduce over "this rank can": a rank whose environment says otherwise cannot
leave the others waiting in a send/recv), it is skipped on more ranks than local devices (the MIN over ranks would measure
the NIC, not xGMI), and any failure falls back to the constant.  USP_LINK_GBS=<GB/s> pins the figure.

The same opt-in moment measures the second figure `_link_bound` compares the link with: the rate of the forward attention
kernel on a launch large enough to fill the part (B1 S4096 H16 D128 causal, five launches), again reduced with MIN over
the ranks.  Without the probe it is the constant 1.1e15 FLOP/s (profiles/: 1.10-1.17 PFLOP/s at the C2 shape); USP_KERNEL_TFS
pins it.  `device_cus()` is where the head-group sizing gets its "work items per CU" unit from (the device's CU count; 256
when no device is visible)."""
import os

import torch
import torch.distributed as dist

DEFAULT_BYTES_PER_S = 64e9
DEFAULT_KERNEL_FLOPS_PER_S = 1.1e15
_measured = None          # bytes/s, one link, one direction
_kernel_measured = None   # FLOP/s of the forward kernel on a part-filling launch
_cus = None


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


def kernel_flops_per_s() -> float:
    env = os.environ.get("USP_KERNEL_TFS")
    if env:
        try:
            return float(env) * 1e12
        except ValueError:
            pass
    return _kernel_measured if _kernel_measured else DEFAULT_KERNEL_FLOPS_PER_S


def kernel_measured() -> bool:
    return _kernel_measured is not None


def device_cus() -> int:
    """Compute units of the current device (the unit of the head-group sizing: work items per CU); 256 without a device."""
    global _cus
    if _cus is None:
        try:
            _cus = int(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count) \
                if torch.cuda.is_available() else 256
        except Exception:
            _cus = 256
    return _cus


def _probe_kernel_rate(dev):
    """FLOP/s of the forward kernel on B1 S4096 H16 D128 causal bf16 (68.7 GFLOP per launch), device events."""
    from .. import _C
    B, S, H, D = 1, 4096, 16, 128
    g = torch.Generator(device=dev)            # a private generator: the probe must not advance the user's RNG stream
    g.manual_seed(0x5eed)                      # (N(0,1) operands on purpose: the kernel's rate depends on the data it multiplies)
    q, k, v = (torch.randn(B, S, H, D, device=dev, dtype=torch.float32, generator=g).to(torch.bfloat16) for _ in range(3))
    out = torch.empty_like(q)
    lse = torch.empty(B, H, S, device=dev, dtype=torch.float32)
    for _ in range(3):
        _C.flash_fwd(q, k, v, D ** -0.5, True, lse, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    n = 5
    e0.record()
    for _ in range(n):
        _C.flash_fwd(q, k, v, D ** -0.5, True, lse, out=out)
    e1.record()
    e1.synchronize()
    return n * 4.0 * B * H * S * S * D * 0.5 / max(e0.elapsed_time(e1) * 1e-3, 1e-9)


def probe_link_rate(rank: int, world_size: int, nbytes: int = 16 << 20, rounds: int = 4):
    """Collective over the default group; opt-in (USP_LINK_PROBE=1).  Only on RCCL with more than one rank, one device per
    rank on ONE node; anywhere else (gloo tests, one rank, several nodes) the constant stays.  Returns bytes/s or None."""
    global _measured
    if os.environ.get("USP_LINK_PROBE", "0") != "1":
        return None
    if world_size < 2 or not dist.is_initialized() or dist.get_backend() != "nccl" or not torch.cuda.is_available():
        return None
    try:
        dev = torch.device("cuda", torch.cuda.current_device())
        can = (not os.environ.get("USP_LINK_GBS")) and world_size <= torch.cuda.device_count()
        agree = torch.tensor([1 if can else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)        # every rank runs the probe, or none does
        if int(agree.item()) == 0:
            return None
        # From here on every rank has agreed to probe: whatever fails locally, the rank still takes part in the two
        # remaining all-reduces with a sentinel (0 = "no figure"), so a failure here costs the figure, not the job.
        local_rate = 0.0
        try:
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
            local_rate = rounds * nbytes / max(e0.elapsed_time(e1) * 1e-3, 1e-9)
        except Exception as e:
            import warnings
            warnings.warn(f"usp link probe: the send/recv rounds failed on rank {rank} ({e!r}); no figure")
        rate = torch.tensor([local_rate], dtype=torch.float64, device=dev)
        dist.all_reduce(rate, op=dist.ReduceOp.MIN)         # one figure for every rank; 0 from any rank = none
        _measured = float(rate.item()) or None
        if not os.environ.get("USP_KERNEL_TFS"):
            global _kernel_measured
            try:
                kr = _probe_kernel_rate(dev)
            except Exception:
                kr = 0.0
            krt = torch.tensor([kr], dtype=torch.float64, device=dev)
            dist.all_reduce(krt, op=dist.ReduceOp.MIN)       # every rank reaches this all-reduce, with 0 if its probe failed
            if float(krt.item()) > 0:
                _kernel_measured = float(krt.item())
        return _measured
    except Exception as e:                                   # a failed probe must not take the set-up down
        import warnings
        warnings.warn(f"usp link probe failed ({e!r}); the constant {DEFAULT_BYTES_PER_S / 1e9:.0f} GB/s stays")
        return None
