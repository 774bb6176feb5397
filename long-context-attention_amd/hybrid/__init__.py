from .attn_layer import LongContextAttention
from .utils import RING_IMPL_DICT, RING_IMPL_QKVPACKED_DICT

__all__ = ["LongContextAttention", "RING_IMPL_DICT", "RING_IMPL_QKVPACKED_DICT"]
