from .ring_flash_attn import (
    ring_flash_attn_func,
    ring_flash_attn_kvpacked_func,
    ring_flash_attn_qkvpacked_func,
)
from .zigzag_ring_flash_attn import (
    zigzag_ring_flash_attn_func,
    zigzag_ring_flash_attn_kvpacked_func,
    zigzag_ring_flash_attn_qkvpacked_func,
)
from .stripe_flash_attn import (
    stripe_flash_attn_func,
    stripe_flash_attn_kvpacked_func,
    stripe_flash_attn_qkvpacked_func,
)
from .ring_flash_attn_varlen import (
    ring_flash_attn_varlen_func,
    ring_flash_attn_varlen_kvpacked_func,
    ring_flash_attn_varlen_qkvpacked_func,
)
from .zigzag_ring_flash_attn_varlen import (
    zigzag_ring_flash_attn_varlen_func,
    zigzag_ring_flash_attn_varlen_kvpacked_func,
    zigzag_ring_flash_attn_varlen_qkvpacked_func,
)
from .varlen_utils import extract_local_varlen, flatten_lse, unflatten_lse
from .utils import RingComm, KVRelay, update_out_and_lse

__all__ = [
    "ring_flash_attn_func", "ring_flash_attn_kvpacked_func", "ring_flash_attn_qkvpacked_func",
    "zigzag_ring_flash_attn_func", "zigzag_ring_flash_attn_kvpacked_func",
    "zigzag_ring_flash_attn_qkvpacked_func", "stripe_flash_attn_func",
    "stripe_flash_attn_kvpacked_func", "stripe_flash_attn_qkvpacked_func", "RingComm", "KVRelay",
    "update_out_and_lse", "ring_flash_attn_varlen_func", "ring_flash_attn_varlen_kvpacked_func",
    "ring_flash_attn_varlen_qkvpacked_func", "zigzag_ring_flash_attn_varlen_func",
    "zigzag_ring_flash_attn_varlen_kvpacked_func", "zigzag_ring_flash_attn_varlen_qkvpacked_func",
    "extract_local_varlen", "flatten_lse", "unflatten_lse",
]
