"""The sequence-parallel attention layers (yunchang.hybrid): the sequential layer, its packed-qkv form, the
head-group pipelined layer, and the name -> ring function registries."""
from . import async_attn_layer, attn_layer, utils

_EXPORTS = {attn_layer: ("LongContextAttention", "LongContextAttentionQKVPacked"),
            async_attn_layer: ("AsyncLongContextAttention",),
            utils: ("RING_IMPL_DICT", "RING_IMPL_QKVPACKED_DICT")}
__all__ = []
for _module, _names in _EXPORTS.items():
    for _name in _names:
        globals()[_name] = getattr(_module, _name)
        __all__.append(_name)
