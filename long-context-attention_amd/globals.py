"""Process-grid state of the USP path: same surface as yunchang/globals.py.

`set_seq_parallel_pg` builds the 2-D ulysses x ring (x data-parallel) grid exactly like
yunchang/globals.py:22-81: with use_ulysses_low the ulysses ranks are contiguous and ring ranks are
strided by the ulysses degree.  For one 8 x MI355X node every pair of GPUs is a direct xGMI
neighbour, so both the ulysses pairs {2i, 2i+1} and the stride-2 ring are single-hop.
"""
import torch
import torch.distributed as dist


class Singleton:
    _instance = None

    def __new__(cls, *args, **kwargs):
        if not cls._instance:
            cls._instance = super(Singleton, cls).__new__(cls, *args, **kwargs)
        return cls._instance


class ProcessGroupSingleton(Singleton):
    def __init__(self):
        self.ULYSSES_PG = None
        self.RING_PG = None


PROCESS_GROUP = ProcessGroupSingleton()


def set_seq_parallel_pg(sp_ulysses_degree, sp_ring_degree, rank, world_size, use_ulysses_low=True):
    """sp_ulysses_degree x sp_ring_degree = seq_parallel_degree; world_size // that = dp degree.
    Every rank must call this (every rank creates every group, globals.py:49,55,68,76)."""
    sp_degree = sp_ring_degree * sp_ulysses_degree
    dp_degree = world_size // sp_degree
    assert world_size % sp_degree == 0, f"world_size {world_size} % sp_degree {sp_ulysses_degree} == 0"

    num_ulysses_pgs = sp_ring_degree
    num_ring_pgs = sp_ulysses_degree
    ulysses_pg = ring_pg = None
    for dp_rank in range(dp_degree):
        offset = dp_rank * sp_degree
        if use_ulysses_low:
            ulysses_lists = [list(range(i * sp_ulysses_degree + offset, (i + 1) * sp_ulysses_degree + offset))
                             for i in range(num_ulysses_pgs)]
            ring_lists = [list(range(i + offset, sp_degree + offset, num_ring_pgs))
                          for i in range(num_ring_pgs)]
            order = [("u", r) for r in ulysses_lists] + [("r", r) for r in ring_lists]
        else:
            ring_lists = [list(range(i * sp_ring_degree + offset, (i + 1) * sp_ring_degree + offset))
                          for i in range(num_ring_pgs)]
            ulysses_lists = [list(range(i + offset, sp_degree + offset, num_ulysses_pgs))
                             for i in range(num_ulysses_pgs)]
            order = [("r", r) for r in ring_lists] + [("u", r) for r in ulysses_lists]
        for kind, ranks in order:
            group = dist.new_group(ranks)
            if rank in ranks:
                if kind == "u":
                    ulysses_pg = group
                else:
                    ring_pg = group

    PROCESS_GROUP.ULYSSES_PG = ulysses_pg
    PROCESS_GROUP.RING_PG = ring_pg
    from .comm import relay_exchange
    relay_exchange.GRID = (sp_ulysses_degree, sp_ring_degree, world_size, bool(use_ulysses_low))      # who is whose peer
    relay_exchange.forget_agreements()             # buffer signatures confirmed on the previous grid mean nothing on this one
    # every rank is here by contract: the one collective moment to measure what the schedules size themselves by --
    # opt-in (USP_LINK_PROBE=1, comm/link.py): a reference-style script must not meet a hidden collective here
    from .comm.link import probe_link_rate
    probe_link_rate(rank, world_size)
    from .ring.utils import forget_groups
    forget_groups()                 # (size, rank) cache of process groups: a re-initialised grid starts empty


# Feature flags of the reference (globals.py:83-135), kept so `from yunchang.globals import HAS_*`
# keeps importing.  None of the third-party kernels is used by this package.
HAS_FLASH_ATTN = False
HAS_FLASH_ATTN_HOPPER = False
HAS_FLASHINFER = False
HAS_AITER = False
HAS_SAGE_ATTENTION = False
HAS_SPARSE_SAGE_ATTENTION = False
HAS_NPU = False


def has_usp_hip() -> bool:
    """True when libusp_hip.so is built and loadable."""
    from . import _C
    try:
        _C.load()
        return True
    except (RuntimeError, OSError):
        return False
