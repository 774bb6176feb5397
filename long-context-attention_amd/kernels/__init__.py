"""Kernel selector of the USP path: same surface as yunchang/kernels/__init__.py:38-295.

`AttnType` keeps every member (and string value) of the reference enum (:38-53) so existing
callers and `AttnType.from_string` (:55-60) keep working, plus `HIP` for this package's kernel.
On this package's path every dense-attention type is served by the hand-written gfx950 kernel;
the vendor-specific types that are out of scope (SageAttention, FlashInfer, Ascend NPU) raise.
"""
from enum import Enum

from .attention import (
    HipBlockBackend,
    get_block_backend,
    hip_attn_backward,
    hip_attn_forward,
    hip_attn_func,
    set_block_backend,
)


class AttnType(Enum):
    AITER = "aiter"
    FA = "fa"
    FA3 = "fa3"
    FLASHINFER = "flashinfer"
    TORCH_MATH = "torch_math"
    TORCH_FLASH = "torch_flash"
    TORCH_EFFICIENT = "torch_efficient"
    TORCH_CUDNN = "torch_cudnn"
    SAGE_AUTO = "sage_auto"
    SAGE_FP16 = "sage_fp16"
    SAGE_FP16_TRITON = "sage_fp16_triton"
    SAGE_FP8 = "sage_fp8"
    SAGE_FP8_SM90 = "sage_fp8_sm90"
    SPARSE_SAGE = "sparse_sage"
    NPU = "npu"
    HIP = "hip"          # the MI355X-native kernel of this package

    @classmethod
    def from_string(cls, s: str):
        for member in cls:
            if member.value == s:
                return member
        raise ValueError(f"'{s}' is not a valid {cls.__name__}")


# Dense softmax attention in 16-bit with fp32 accumulation: all served by the HIP kernel.
_DENSE = {AttnType.HIP, AttnType.FA, AttnType.FA3, AttnType.AITER, AttnType.TORCH_MATH,
          AttnType.TORCH_FLASH, AttnType.TORCH_EFFICIENT, AttnType.TORCH_CUDNN}


def select_flash_attn_impl(impl_type: AttnType, stage: str = "fwd-bwd", attn_processor=None):
    """kernels/__init__.py:63-295.  Stages: "fwd-only" -> (out, lse); "bwd-only" -> in-place
    dq/dk/dv; "fwd-bwd" -> autograd-aware function."""
    if impl_type in _DENSE:
        if stage == "fwd-only":
            return hip_attn_forward
        if stage == "bwd-only":
            return hip_attn_backward
        if stage == "fwd-bwd":
            return hip_attn_func
        raise ValueError(f"Unknown stage: {stage}")
    if attn_processor is not None:          # reference escape hatch, kernels/__init__.py:292-293
        return attn_processor
    if isinstance(impl_type, AttnType):
        raise ValueError(
            f"AttnType.{impl_type.name} is a third-party backend outside the scope of the MI355X USP "
            f"path; use AttnType.HIP (or any dense type, which maps to it)")
    raise ValueError(f"Unknown flash attention implementation: {impl_type}")


__all__ = ["AttnType", "select_flash_attn_impl", "hip_attn_forward", "hip_attn_backward",
           "hip_attn_func", "HipBlockBackend", "get_block_backend", "set_block_backend"]
