"""Ring attention schedules (yunchang.ring): contiguous ("basic"), zigzag and stripe layouts, dense and packed
variable-length, each as *_func / *_kvpacked_func / *_qkvpacked_func, plus the ring plumbing."""
from . import (ring_flash_attn, ring_flash_attn_varlen, stripe_flash_attn, utils, varlen_utils,
               zigzag_ring_flash_attn, zigzag_ring_flash_attn_varlen)

__all__ = []


def _export(module, names):
    for name in names:
        globals()[name] = getattr(module, name)
        __all__.append(name)


for _module, _stem in ((ring_flash_attn, "ring_flash_attn"), (zigzag_ring_flash_attn, "zigzag_ring_flash_attn"),
                       (stripe_flash_attn, "stripe_flash_attn"), (ring_flash_attn_varlen, "ring_flash_attn_varlen"),
                       (zigzag_ring_flash_attn_varlen, "zigzag_ring_flash_attn_varlen")):
    _export(_module, [f"{_stem}{suffix}" for suffix in ("_func", "_kvpacked_func", "_qkvpacked_func")])
_export(utils, ["RingComm", "KVRelay", "update_out_and_lse"])
_export(varlen_utils, ["extract_local_varlen", "flatten_lse", "unflatten_lse"])
