"""Global -> local shard layouts: same surface as yunchang/comm/extract_local.py.

These run in test / benchmark set-up only (never inside the timed path), so they are plain
tensor indexing.  `zigzag`: 2*rd chunks; ring rank r owns chunks r and 2rd-1-r (so every ring step
does the same amount of causal work), concatenated, then split ud ways by ulysses rank
(extract_local.py:29-49).  `basic`: contiguous chunk (:25-26).  `strip`: token-interleaved (:7-22).
"""
import torch
import torch.distributed as dist

from ..globals import PROCESS_GROUP


def stripe_extract_local(value, rank, world_size, rd, ud, *args, **kwargs):
    assert value.dim() >= 2
    batch_size, seqlen, *rest = value.shape
    assert dist.get_world_size(group=PROCESS_GROUP.RING_PG) == rd
    assert dist.get_world_size(group=PROCESS_GROUP.ULYSSES_PG) == ud
    # token t goes to ring rank t % rd: (B, S/rd, rd, ...) -> (B, rd, S/rd, ...)
    value = value.reshape(batch_size, seqlen // rd, rd, -1).transpose(1, 2)
    value = value.reshape(batch_size, seqlen, -1).chunk(world_size, dim=1)[rank]
    return value.reshape([batch_size, seqlen // world_size] + rest).contiguous()


def basic_extract_local(value, rank, world_size, *args, **kwargs):
    return value.chunk(world_size, dim=1)[rank].detach().clone()


def zigzag_extract_local(value, rank, world_size, rd, ud, dim=1, *args, **kwargs):
    """value (bs, seqlen, ...) -> this rank's (bs, seqlen / world_size, ...) shard."""
    assert value.dim() >= 2
    batch_size, seqlen, *rest = value.shape
    r_rank = dist.get_rank(group=PROCESS_GROUP.RING_PG)
    u_rank = dist.get_rank(group=PROCESS_GROUP.ULYSSES_PG)
    assert dist.get_world_size(group=PROCESS_GROUP.RING_PG) == rd
    assert dist.get_world_size(group=PROCESS_GROUP.ULYSSES_PG) == ud
    chunks = value.chunk(2 * rd, dim=dim)
    local = torch.cat([chunks[r_rank], chunks[2 * rd - r_rank - 1]], dim=dim).chunk(ud, dim=dim)[u_rank]
    return local.reshape([batch_size, seqlen // world_size] + rest).contiguous()


EXTRACT_FUNC_DICT = {
    "basic": basic_extract_local,
    "strip": stripe_extract_local,
    "zigzag": zigzag_extract_local,
    "basic_pytorch": basic_extract_local,
    "basic_flashinfer": basic_extract_local,
    "basic_npu": basic_extract_local,
}
