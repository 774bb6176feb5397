"""Shared pieces of the packed variable-length ring schedules.

The reference moves between three LSE layouts (padded (num_seq,H,max_seqlen) out of flash-attn,
flattened (H,T) for the merge, (T,H,1) inside update_out_and_lse; yunchang/ring/utils.py:96-117,
ring/triton_utils.py) and gathers half sequences with boolean masks
(zigzag_ring_flash_attn_varlen.py:27-58,127-140).  Here every kernel call addresses its rows through
two small int32 tables of (first_row, rows) per sequence, built once per call ON THE DEVICE from
cu_seqlens (no host synchronisation), the LSE lives in the flattened (H,T) layout throughout, and the
padded layout is only materialised when the caller asks for the reference's return value.
"""
import torch


class SeqTables:
    """(num_seq, 2) int32 device tables of (first_row, rows): whole sequences, front halves, back halves."""

    def __init__(self, cu_seqlens: torch.Tensor, max_seqlen: int, device):
        cu = cu_seqlens.to(device=device, dtype=torch.int32)
        first, nxt = cu[:-1], cu[1:]
        length = nxt - first
        half = length // 2
        self.n = int(cu.shape[0]) - 1
        self.max_full = int(max_seqlen)
        self.max_half = (int(max_seqlen) + 1) // 2
        self.full = torch.stack([first, length], 1).contiguous()
        self.front = torch.stack([first, half], 1).contiguous()
        self.back = torch.stack([first + half, length - half], 1).contiguous()


def unflatten_lse(lse_flat: torch.Tensor, cu_seqlens: torch.Tensor, max_seqlen: int) -> torch.Tensor:
    """(H,T) -> the reference's padded (num_seq, H, max_seqlen) (unflatten_varlen_lse, ring/utils.py:105-117;
    padding is zero here, uninitialised there)."""
    H, T = lse_flat.shape
    cu = cu_seqlens.to(device=lse_flat.device, dtype=torch.int64)
    n = cu.shape[0] - 1
    length = cu[1:] - cu[:-1]
    seq_id = torch.repeat_interleave(torch.arange(n, device=lse_flat.device), length, output_size=T)
    pos = torch.arange(T, device=lse_flat.device) - cu[:-1][seq_id]
    out = torch.zeros((n, H, int(max_seqlen)), dtype=lse_flat.dtype, device=lse_flat.device)
    out[seq_id, :, pos] = lse_flat.t()
    return out


def flatten_lse(lse_padded: torch.Tensor, cu_seqlens: torch.Tensor) -> torch.Tensor:
    """The reference's padded (num_seq, H, max_seqlen) -> (H,T) (flatten_varlen_lse, ring/utils.py:96-103)."""
    n, H, _ = lse_padded.shape
    cu = cu_seqlens.to(device=lse_padded.device, dtype=torch.int64)
    T = int(cu[-1])
    length = cu[1:] - cu[:-1]
    seq_id = torch.repeat_interleave(torch.arange(n, device=lse_padded.device), length, output_size=T)
    pos = torch.arange(T, device=lse_padded.device) - cu[:-1][seq_id]
    return lse_padded[seq_id, :, pos].t().contiguous()


def extract_local_varlen(value: torch.Tensor, cu_seqlens, rank: int, world_size: int, layout="zigzag"):
    """Shard a packed (T,...) tensor for ring rank `rank`: per sequence, chunks `rank` and `2P-1-rank` of
    2P (zigzag) or chunk `rank` of P (basic).  The local cu_seqlens are cu_seqlens // world_size.
    (yunchang has no packed-batch extractor; this is the per-sequence form of comm/extract_local.py.)"""
    cu = [int(c) for c in cu_seqlens]
    parts = []
    for a, b in zip(cu[:-1], cu[1:]):
        seq = value[a:b]
        if layout == "zigzag":
            assert (b - a) % (2 * world_size) == 0, "sequence length must be a multiple of 2*world_size"
            ch = seq.chunk(2 * world_size, dim=0)
            parts += [ch[rank], ch[2 * world_size - 1 - rank]]
        else:
            assert (b - a) % world_size == 0, "sequence length must be a multiple of world_size"
            parts.append(seq.chunk(world_size, dim=0)[rank])
    return torch.cat(parts, dim=0).contiguous()
