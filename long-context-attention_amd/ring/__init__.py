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


# Names the reference's yunchang.ring also exports (ring/__init__.py:32-44).  `ring_pytorch_attn_func` is the basic
# ring with a TORCH_* block kernel (ring_pytorch_attn.py): every dense kernel is the HIP kernel here, so it IS
# ring_flash_attn_func (hybrid/utils.py maps "basic_pytorch" the same way).  The FlashInfer / Ascend-NPU rings are
# vendor back ends outside the MI355X path: importable, and they say so when called.
ring_pytorch_attn_func = ring_flash_attn_func          # noqa: F821  (exported above)
__all__.append("ring_pytorch_attn_func")


def _out_of_scope(name):
    def fn(*args, **kwargs):
        raise NotImplementedError(f"{name} is a vendor-specific ring outside the MI355X USP path; use "
                                  f"ring_flash_attn_func / zigzag_ring_flash_attn_func")
    fn.__name__ = name
    return fn


for _name in ("ring_flashinfer_attn_func", "ring_flashinfer_attn_kvpacked_func", "ring_flashinfer_attn_qkvpacked_func",
              "ring_npu_flash_attn_func"):
    globals()[_name] = _out_of_scope(_name)
    __all__.append(_name)
