"""String -> ring function registries: same keys as yunchang/hybrid/utils.py:14-28 (note the key is
"strip", not "stripe").  Variants outside the scope of this package raise when called."""
from ..ring import (
    ring_flash_attn_func,
    ring_flash_attn_qkvpacked_func,
    stripe_flash_attn_func,
    stripe_flash_attn_qkvpacked_func,
    zigzag_ring_flash_attn_func,
    zigzag_ring_flash_attn_qkvpacked_func,
)


def _out_of_scope(name):
    def fn(*args, **kwargs):
        raise NotImplementedError(
            f"ring_impl_type '{name}' is not part of the MI355X USP path; use 'zigzag' (causal, load "
            f"balanced) or 'basic'")
    fn.__name__ = f"{name}_not_in_scope"
    return fn


RING_IMPL_DICT = {
    "basic": ring_flash_attn_func,
    "zigzag": zigzag_ring_flash_attn_func,
    "strip": stripe_flash_attn_func,
    "basic_pytorch": ring_flash_attn_func,
    "basic_flashinfer": _out_of_scope("basic_flashinfer"),
    "basic_npu": _out_of_scope("basic_npu"),
}

RING_IMPL_QKVPACKED_DICT = {
    "basic": ring_flash_attn_qkvpacked_func,
    "zigzag": zigzag_ring_flash_attn_qkvpacked_func,
    "strip": stripe_flash_attn_qkvpacked_func,
    "basic_flashinfer": _out_of_scope("basic_flashinfer"),
}
