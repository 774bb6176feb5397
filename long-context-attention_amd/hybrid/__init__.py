from .attn_layer import LongContextAttention, LongContextAttentionQKVPacked
from .async_attn_layer import AsyncLongContextAttention
from .utils import RING_IMPL_DICT, RING_IMPL_QKVPACKED_DICT

__all__ = ["LongContextAttention", "LongContextAttentionQKVPacked", "AsyncLongContextAttention",
           "RING_IMPL_DICT", "RING_IMPL_QKVPACKED_DICT"]
