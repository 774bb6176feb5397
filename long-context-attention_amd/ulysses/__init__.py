"""Pure head-parallel attention (yunchang.ulysses)."""
from . import attn_layer as _layer

UlyssesAttention = _layer.UlyssesAttention
__all__ = ["UlyssesAttention"]
