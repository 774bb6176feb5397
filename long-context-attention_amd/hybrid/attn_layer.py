"""LongContextAttention: same surface as yunchang/hybrid/attn_layer.py:14-161.

forward = Ulysses head all-to-all of q, k, v (attn_layer.py:111-119) -> ring attention over the
ring group (:132-147) -> all-to-all of the output back to sequence sharding (:156-158).

MI355X-first: q, k and v travel in ONE packed head all-to-all (GQA-capable), and the exchanges are pipelined over
head groups on a side HIP stream behind the attention kernels instead of running exposed in front of and behind
them (LongContextAttention._packed_exchange); results are identical.
"""
import os
from typing import Any

import torch
import torch.distributed as dist
from torch import Tensor

from ..comm.all_to_all import SeqAllToAll4D, SeqAllToAll5D
from ..globals import PROCESS_GROUP
from ..kernels import AttnType
from ..kernels.attention import kernel_head_dim, pad_head_dim, window_of
from ..ring.zigzag_ring_flash_attn import _check_hot_path_args
from .async_attn_layer import _AsyncUSPFunc, _MAX_GROUPS, _RING_FWD_BWD, pipeline_mode
from .utils import RING_IMPL_DICT, RING_IMPL_QKVPACKED_DICT


def _first(result):
    """Ring functions return `out` or `(out, lse, None)`; the layers only hand `out` on."""
    return result[0] if isinstance(result, tuple) else result


class _USPLayer(torch.nn.Module):
    """What both layers share: the 2-D process grid of set_seq_parallel_pg and the two exchanges."""

    def __init__(self, scatter_idx: int, gather_idx: int, use_sync: bool, attn_type: AttnType) -> None:
        super().__init__()
        self.ring_pg, self.ulysses_pg = PROCESS_GROUP.RING_PG, PROCESS_GROUP.ULYSSES_PG
        assert (
            self.ulysses_pg is not None or self.ring_pg is not None
        ), f"use set_seq_parallel_pg() first. Now ulysses pg {self.ulysses_pg} and ring pg {self.ring_pg}"
        self.scatter_idx, self.gather_idx = scatter_idx, gather_idx
        self.use_sync, self.attn_type = use_sync, attn_type
        self._ulysses_size = self._ring_size = None

    @property
    def ulysses_size(self) -> int:
        """Ulysses degree; the grid is fixed once the groups exist, so torch.distributed is asked once (on the
        first forward -- the constructor, like the reference's, only checks that a group is set)."""
        if self._ulysses_size is None:
            self._ulysses_size = dist.get_world_size(self.ulysses_pg)
        return self._ulysses_size

    @property
    def ring_size(self) -> int:
        if self._ring_size is None:
            self._ring_size = dist.get_world_size(self.ring_pg) if self.ring_pg is not None else 1
        return self._ring_size

    def _ring_options(self, dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes, deterministic,
                      return_attn_probs):
        return dict(dropout_p=dropout_p, softmax_scale=softmax_scale, causal=causal, window_size=window_size,
                    softcap=softcap, alibi_slopes=alibi_slopes, deterministic=deterministic,
                    return_attn_probs=return_attn_probs, group=self.ring_pg, attn_type=self.attn_type)


class LongContextAttention(_USPLayer):
    """Unified sequence parallel attention (ulysses x ring), arguments as in the reference:
        scatter_idx / gather_idx (int): dims of the all-to-all (2 = heads, 1 = sequence)
        ring_impl_type (str): key of RING_IMPL_DICT ("basic" | "zigzag" | "strip")
        use_pack_qkv (bool): q, k and v travel in ONE head all-to-all instead of three.  This layer ALWAYS
            packs (GQA-capable buffer (P, S/P, B, hq + 2 hkv, D), hybrid/async_attn_layer.py:_qkv_to_seq) --
            the flag is accepted and changes nothing; results are identical either way.  (The reference's
            packed branch is dead code: `.continous()` typo at attn_layer.py:88, and it cannot express GQA.)
            USP_PACK_QKV=0 restores the reference's three-exchange structure (an A/B switch for multi-GPU
            measurements).
        use_sync (bool): synchronise the device after each all-to-all (three blocking exchanges, as the
            reference)
        attn_type (AttnType): any dense type; all are served by the gfx950 kernel
    Beside packing, the exchange is PIPELINED over head groups on a side HIP stream behind the attention
    kernels whenever there is more than one group to pipeline (hybrid/async_attn_layer.py:pipeline_mode:
    USP_PIPELINE_ULYSSES=0 disables it, =1 enables it also beside a ring, USP_SAFE_COMM=1 keeps ONE communicator
    in flight at any time).
    """

    def __init__(self, scatter_idx: int = 2, gather_idx: int = 1, ring_impl_type: str = "basic",
                 use_pack_qkv: bool = False, use_sync: bool = False,
                 attn_type: AttnType = AttnType.FA, attn_processor: torch.nn.Module = None) -> None:
        super().__init__(scatter_idx, gather_idx, use_sync, attn_type)
        self.use_pack_qkv = use_pack_qkv
        self.attn_processor = attn_processor
        self.ring_attn_fn = RING_IMPL_DICT[ring_impl_type]
        self.ring_impl_type = ring_impl_type

    def _packed_exchange(self, query: Tensor, key: Tensor):
        """None: exchange q, k, v separately (the reference's structure); otherwise the cap on the number of
        head groups of the packed exchange (1 = one packed exchange in front of the attention and one behind
        it; more = pipelined over head groups on the side stream, hybrid/async_attn_layer.py -- identical
        results).  See `pipeline_mode` for when the exchange is pipelined."""
        if self.ulysses_size == 1:
            return None
        if os.environ.get("USP_PACK_QKV", "1") == "0" or self.use_sync or self.attn_processor is not None:
            return None
        if (self.scatter_idx, self.gather_idx) != (2, 1) or self.ring_impl_type not in _RING_FWD_BWD:
            return None
        P = self.ulysses_size
        if query.shape[2] % P or key.shape[2] % P:
            return None
        return _MAX_GROUPS if pipeline_mode(self.ring_size) else 1

    def forward(self, query: Tensor, key: Tensor, value: Tensor, dropout_p=0.0, softmax_scale=None,
                causal=False, window_size=(-1, -1), softcap=0.0, alibi_slopes=None,
                deterministic=False, return_attn_probs=False, *args: Any) -> Tensor:
        """query (bs, seq_len/N, head_cnt, head_size); key/value (bs, seq_len/N, kv_head_cnt,
        head_size) -> context (bs, seq_len/N, head_cnt, head_size)."""
        D = query.shape[-1]
        if kernel_head_dim(D) != D:      # a head dim the kernels do not instantiate (e.g. 96): zero-padded copies
            out = self.forward(*pad_head_dim(query, key, value), dropout_p, D ** -0.5 if softmax_scale is None else softmax_scale,
                               causal, window_size, softcap, alibi_slopes, deterministic, return_attn_probs, *args)
            return out[..., :D]
        ng_cap = self._packed_exchange(query, key)
        if ng_cap is not None and window_of(window_size) is not None:
            ng_cap = None         # a sliding window: the reference's structure (three exchanges); the ring function
                                  # serves it at ring degree 1 and refuses beyond (ring/ring_flash_attn.py)
        if ng_cap is not None:
            assert alibi_slopes is None
            _check_hot_path_args(dropout_p, window_size, softcap)
            return _AsyncUSPFunc.apply(query, key, value, softmax_scale, causal, self.ulysses_pg, self.ring_pg,
                                       self.ring_impl_type, ng_cap)
        options = self._ring_options(dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes,
                                     deterministic, return_attn_probs)
        if self.ulysses_size == 1:      # nothing to exchange (the reference still makes 8 layout copies here)
            return _first(self.ring_attn_fn(query, key, value, attn_processor=self.attn_processor, **options))
        # sequence shards -> head shards: (bs, seq_len/N, heads, d) -> (bs, seq_len, heads/N, d)
        q, k, v = (SeqAllToAll4D.apply(self.ulysses_pg, t, self.scatter_idx, self.gather_idx, self.use_sync, False)
                   for t in (query, key, value))
        context = _first(self.ring_attn_fn(q, k, v, attn_processor=self.attn_processor, **options))
        # ... and back: (bs, seq_len, heads/N, d) -> (bs, seq_len/N, heads, d)
        return SeqAllToAll4D.apply(self.ulysses_pg, context, self.gather_idx, self.scatter_idx, self.use_sync, False)


class LongContextAttentionQKVPacked(_USPLayer):
    """Same surface as yunchang/hybrid/attn_layer.py:164-259: packed
    qkv (bs, seq_len/N, 3, head_cnt, head_size) -> ONE head all-to-all (instead of three) -> ring
    attention on the packed views -> all-to-all of the output back.  Equal head counts only
    (a packed tensor cannot express GQA, README.md:193)."""

    def __init__(self, scatter_idx: int = 3, gather_idx: int = 1, ring_impl_type: str = "basic",
                 use_sync: bool = False, attn_type: AttnType = AttnType.FA) -> None:
        super().__init__(scatter_idx, gather_idx, use_sync, attn_type)
        self.ring_attn_fn = RING_IMPL_QKVPACKED_DICT[ring_impl_type]

    def forward(self, qkv, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1),
                softcap=0.0, alibi_slopes=None, deterministic=False, return_attn_probs=False,
                *args: Any) -> Tensor:
        D = qkv.shape[-1]
        if kernel_head_dim(D) != D:
            out = self.forward(pad_head_dim(qkv)[0], dropout_p, D ** -0.5 if softmax_scale is None else softmax_scale, causal,
                               window_size, softcap, alibi_slopes, deterministic, return_attn_probs, *args)
            return out[..., :D]
        exchange = self.ulysses_size > 1
        if exchange:         # scatter 3 (heads), gather 1 (sequence)
            qkv = SeqAllToAll5D.apply(self.ulysses_pg, qkv, self.scatter_idx, self.gather_idx, self.use_sync, False)
        out = _first(self.ring_attn_fn(qkv, **self._ring_options(dropout_p, softmax_scale, causal, window_size,
                                                                  softcap, alibi_slopes, deterministic,
                                                                  return_attn_probs)))
        if exchange:         # (bs, seq_len, head_cnt/N, head_size) -> (bs, seq_len/N, head_cnt, head_size)
            out = SeqAllToAll4D.apply(self.ulysses_pg, out, self.gather_idx, self.scatter_idx - 1,
                                      self.use_sync, False)
        return out
