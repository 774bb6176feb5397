"""Global -> local shard layouts: same surface as yunchang/comm/extract_local.py (names, argument order, the
EXTRACT_FUNC_DICT keys).

These run in test / benchmark set-up only (never inside the timed path).  Each layout is expressed as the list
of GLOBAL token positions a rank owns, and the shard is one index_select along the sequence:
  basic  (extract_local.py:25-26)  rank r owns the contiguous block r of world_size;
  zigzag (:29-49)  the sequence is cut into 2*rd chunks, ring rank i owns chunks i and 2rd-1-i (every ring step
                   then does the same amount of causal work); the ulysses rank takes its 1/ud of that;
  strip  (:7-22)   token t belongs to ring position t % rd; the ring-major order is cut into world_size blocks.
"""
import torch
import torch.distributed as dist

from ..globals import PROCESS_GROUP


def _grid(rd: int, ud: int):
    """(ring rank, ulysses rank) of this process; the grid must be the one set_seq_parallel_pg built."""
    ring, ulysses = PROCESS_GROUP.RING_PG, PROCESS_GROUP.ULYSSES_PG
    assert dist.get_world_size(group=ring) == rd
    assert dist.get_world_size(group=ulysses) == ud
    return dist.get_rank(group=ring), dist.get_rank(group=ulysses)


def _take(value: torch.Tensor, positions: torch.Tensor, dim: int = 1) -> torch.Tensor:
    return value.index_select(dim, positions.to(value.device)).contiguous()


def basic_extract_local(value, rank, world_size, *args, **kwargs):
    n = value.shape[1] // world_size
    return _take(value.detach(), torch.arange(rank * n, (rank + 1) * n))


def zigzag_extract_local(value, rank, world_size, rd, ud, dim=1, *args, **kwargs):
    """value (bs, seqlen, ...) -> this rank's (bs, seqlen / world_size, ...) shard."""
    assert value.dim() >= 2
    i, u = _grid(rd, ud)
    c = value.shape[dim] // (2 * rd)                       # chunk length
    mine = torch.cat([torch.arange(i * c, (i + 1) * c), torch.arange((2 * rd - 1 - i) * c, (2 * rd - i) * c)])
    n = mine.numel() // ud
    return _take(value, mine[u * n:(u + 1) * n], dim)


def stripe_extract_local(value, rank, world_size, rd, ud, *args, **kwargs):
    assert value.dim() >= 2
    _grid(rd, ud)
    seqlen = value.shape[1]
    ring_major = torch.arange(seqlen).reshape(seqlen // rd, rd).t().reshape(-1)   # positions j, j+rd, ... per j
    n = seqlen // world_size
    return _take(value, ring_major[rank * n:(rank + 1) * n])


_LAYOUTS = {"basic": basic_extract_local, "strip": stripe_extract_local, "zigzag": zigzag_extract_local}
# the reference also registers its backend-specific ring names; they shard like "basic"
EXTRACT_FUNC_DICT = dict(_LAYOUTS, **{f"basic_{suffix}": basic_extract_local
                                      for suffix in ("pytorch", "flashinfer", "npu")})
