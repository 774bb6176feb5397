from .attn_layer import UlyssesAttention

__all__ = ["UlyssesAttention"]
