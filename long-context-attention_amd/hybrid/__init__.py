from .attn_layer import LongContextAttention, LongContextAttentionQKVPacked
from .utils import RING_IMPL_DICT, RING_IMPL_QKVPACKED_DICT

__all__ = ["LongContextAttention", "LongContextAttentionQKVPacked", "RING_IMPL_DICT",
           "RING_IMPL_QKVPACKED_DICT"]
