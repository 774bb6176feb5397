"""UlyssesAttention: same surface as yunchang/ulysses/attn_layer.py:15-126.

Pure head parallelism: all-to-all q, k, v from sequence sharding to head sharding, one local
attention over the full sequence (the `fwd-bwd` stage of the selector = the HIP kernel with autograd),
all-to-all of the output back.  Equal to LongContextAttention with ring degree 1.
"""
from typing import Any

import torch
import torch.distributed as dist
from torch import Tensor

from ..comm.all_to_all import SeqAllToAll4D
from ..kernels import AttnType, select_flash_attn_impl


class UlyssesAttention(torch.nn.Module):
    """Arguments (identical to the reference):
        sequence_process_group (ProcessGroup): sequence parallel process group
        scatter_idx (int): scatter_idx for all2all comm
        gather_idx (int): gather_idx for all2all comm
        use_sync (bool): synchronize after each all-to-all
        attn_type (AttnType): attention type enum
    """

    def __init__(self, sequence_process_group: dist.ProcessGroup = None, scatter_idx: int = 2,
                 gather_idx: int = 1, use_sync: bool = False, attn_type: AttnType = AttnType.FA) -> None:
        super(UlyssesAttention, self).__init__()
        self.spg = sequence_process_group
        self.scatter_idx = scatter_idx
        self.gather_idx = gather_idx
        self.use_sync = use_sync
        self.attn_type = attn_type
        self.attn_fn = select_flash_attn_impl(self.attn_type, stage="fwd-bwd")

    def forward(self, query: Tensor, key: Tensor, value: Tensor, dropout_p=0.0, softmax_scale=None,
                causal=False, window_size=(-1, -1), softcap=0.0, alibi_slopes=None, deterministic=False,
                return_attn_probs=False, *args: Any) -> Tensor:
        # (bs, seq_len/N, head_cnt, head_size) -> (bs, seq_len, head_cnt/N, head_size)
        q = SeqAllToAll4D.apply(self.spg, query, self.scatter_idx, self.gather_idx, self.use_sync)
        k = SeqAllToAll4D.apply(self.spg, key, self.scatter_idx, self.gather_idx, self.use_sync)
        v = SeqAllToAll4D.apply(self.spg, value, self.scatter_idx, self.gather_idx, self.use_sync)
        if softmax_scale is None:
            softmax_scale = q.shape[-1] ** -0.5
        context_layer = self.attn_fn(q, k, v, dropout_p=dropout_p, softmax_scale=softmax_scale,
                                     causal=causal, window_size=window_size, softcap=softcap,
                                     alibi_slopes=alibi_slopes, deterministic=deterministic,
                                     return_attn_probs=return_attn_probs)
        if isinstance(context_layer, tuple):
            context_layer = context_layer[0]
        # (bs, seq_len, head_cnt/N, head_size) -> (bs, seq_len/N, head_cnt, head_size)
        return SeqAllToAll4D.apply(self.spg, context_layer, self.gather_idx, self.scatter_idx,
                                   self.use_sync)
