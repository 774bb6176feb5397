"""UlyssesAttention: same surface as yunchang/ulysses/attn_layer.py:15-126.

Pure head parallelism, i.e. LongContextAttention with ring degree 1: the sequence-sharded q, k, v are exchanged
into head shards, ONE local attention runs over the full sequence (the `fwd-bwd` stage of the kernel selector
= the HIP kernel with autograd), and the result is exchanged back.  The exchanges are SeqAllToAll4D
(comm/all_to_all.py: one pack pass, one RCCL all-to-all, the receive buffer consumed as a view).
"""
from typing import Any

import torch
import torch.distributed as dist
from torch import Tensor

from ..comm.all_to_all import SeqAllToAll4D
from ..kernels import AttnType, select_flash_attn_impl


class UlyssesAttention(torch.nn.Module):
    """Constructor arguments as in the reference: sequence_process_group, scatter_idx (2 = heads), gather_idx
    (1 = sequence), use_sync (synchronise the device after each exchange), attn_type (any dense AttnType)."""

    def __init__(self, sequence_process_group: dist.ProcessGroup = None, scatter_idx: int = 2,
                 gather_idx: int = 1, use_sync: bool = False, attn_type: AttnType = AttnType.FA) -> None:
        super().__init__()
        self.spg, self.use_sync, self.attn_type = sequence_process_group, use_sync, attn_type
        self.scatter_idx, self.gather_idx = scatter_idx, gather_idx
        self.attn_fn = select_flash_attn_impl(attn_type, stage="fwd-bwd")

    def _to_heads(self, x: Tensor) -> Tensor:      # (bs, seq/N, heads, d) -> (bs, seq, heads/N, d)
        return SeqAllToAll4D.apply(self.spg, x, self.scatter_idx, self.gather_idx, self.use_sync, False)

    def _to_seq(self, x: Tensor) -> Tensor:        # (bs, seq, heads/N, d) -> (bs, seq/N, heads, d)
        return SeqAllToAll4D.apply(self.spg, x, self.gather_idx, self.scatter_idx, self.use_sync, False)

    def forward(self, query: Tensor, key: Tensor, value: Tensor, dropout_p=0.0, softmax_scale=None,
                causal=False, window_size=(-1, -1), softcap=0.0, alibi_slopes=None, deterministic=False,
                return_attn_probs=False, *args: Any) -> Tensor:
        q, k, v = (self._to_heads(t) for t in (query, key, value))
        options = dict(dropout_p=dropout_p, causal=causal, window_size=window_size, softcap=softcap,
                       alibi_slopes=alibi_slopes, deterministic=deterministic, return_attn_probs=return_attn_probs,
                       softmax_scale=q.shape[-1] ** -0.5 if softmax_scale is None else softmax_scale)
        attended = self.attn_fn(q, k, v, **options)
        return self._to_seq(attended[0] if isinstance(attended, tuple) else attended)
