"""String -> ring function registries: same keys as yunchang/hybrid/utils.py:14-28 (note the key is "strip",
not "stripe").  Variants outside the scope of this package raise when called."""
from .. import ring as _ring


def _out_of_scope(name):
    def fn(*args, **kwargs):
        raise NotImplementedError(
            f"ring_impl_type '{name}' is not part of the MI355X USP path; use 'zigzag' (causal, load "
            f"balanced) or 'basic'")
    fn.__name__ = f"{name}_not_in_scope"
    return fn


# key -> stem of the implementing functions in yunchang_amd.ring (<stem>_func / <stem>_qkvpacked_func)
_STEMS = {"basic": "ring_flash_attn", "zigzag": "zigzag_ring_flash_attn", "strip": "stripe_flash_attn"}

RING_IMPL_DICT = {key: getattr(_ring, f"{stem}_func") for key, stem in _STEMS.items()}
RING_IMPL_DICT["basic_pytorch"] = RING_IMPL_DICT["basic"]          # every dense backend is the HIP kernel here
RING_IMPL_DICT.update({key: _out_of_scope(key) for key in ("basic_flashinfer", "basic_npu")})

RING_IMPL_QKVPACKED_DICT = {key: getattr(_ring, f"{stem}_qkvpacked_func") for key, stem in _STEMS.items()}
RING_IMPL_QKVPACKED_DICT["basic_flashinfer"] = _out_of_scope("basic_flashinfer")
