"""ctypes binding of libusp_hip.so (C ABI: include/usp_hip.h).

This is the ONLY device backend of the package.  There is no CPU or eager-PyTorch fallback: if the
library is missing, or a tensor is not a CUDA(ROCm) tensor, the call raises.  PyTorch is used for
device memory and streams only; every kernel runs on ``torch.cuda.current_stream()``.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libusp_hip.so")
_lib = None

USP_BF16, USP_FP16 = 0, 1
USP_LAUNCH_INTERLEAVE = 1      # include/usp_hip.h: launch so that collectives on other streams can slip in
USP_ATTN_WINDOW = 2            # the window_left / window_right fields are valid
USP_FORCE_ROW64 = 4            # ABI v6: the call must be served by the one-wave-per-SIMD (64-row) kernel family ...
USP_FORCE_WAVE32 = 8           # ... or by the two-waves-per-SIMD (32 rows per wave) family
USP_BWD_SKIP_DQ = 16           # usp_flash_bwd: only the dK/dV launch ...
USP_BWD_SKIP_DKDV = 32         # ... only the dQ launch
ABI_VERSION = 7
# usp_last_launch_kinds(): bit -> kernel (include/usp_hip.h, USP_KIND_*)
KINDS = {1: "fwd_row64", 2: "fwd_wave8", 4: "fwd_wave4", 8: "fwd_split_merge", 16: "dkdv_row64", 32: "dkdv_wave8",
         64: "dq_row64", 128: "dq_wave8", 256: "reduce_heads", 512: "reduce_cuts"}
_FAMILY_FLAG = {None: 0, "auto": 0, "row64": USP_FORCE_ROW64, "wave32": USP_FORCE_WAVE32}
# In-process default of the `family` argument of flash_fwd / flash_bwd (tests, A/B runs: bench.py times both families
# on the same box).  "auto": the library decides per launch.
_FAMILY_DEFAULT = "auto"


def set_kernel_family(family) -> str:
    """Default kernel family of dense flash launches issued through this binding: "auto" | "row64" | "wave32";
    returns the previous setting.  Per call the `family` argument overrides it."""
    global _FAMILY_DEFAULT
    if family not in _FAMILY_FLAG:
        raise ValueError(f"kernel family must be one of {sorted(k for k in _FAMILY_FLAG if k)}, got {family!r}")
    prev, _FAMILY_DEFAULT = _FAMILY_DEFAULT, (family or "auto")
    return prev


def _family_flag(family) -> int:
    if family is None:
        family = _FAMILY_DEFAULT
    try:
        return _FAMILY_FLAG[family]
    except KeyError:
        raise ValueError(f"kernel family must be one of {sorted(k for k in _FAMILY_FLAG if k)}, got {family!r}") from None


def last_launch_kinds() -> tuple:
    """Names of the kernels the calling thread's last flash_fwd / flash_bwd launched (usp_last_launch_kinds)."""
    m = load().usp_last_launch_kinds()
    return tuple(name for bit, name in KINDS.items() if m & bit)


class UspTensor(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("stride_b", ctypes.c_int64),
                ("stride_s", ctypes.c_int64), ("stride_h", ctypes.c_int64)]


class UspFwdArgs(ctypes.Structure):
    _fields_ = [("dtype", ctypes.c_int32), ("B", ctypes.c_int32), ("Sq", ctypes.c_int32),
                ("Sk", ctypes.c_int32), ("Hq", ctypes.c_int32), ("Hkv", ctypes.c_int32),
                ("D", ctypes.c_int32), ("causal", ctypes.c_int32),
                ("softmax_scale", ctypes.c_float),
                ("q", UspTensor), ("k", UspTensor), ("v", UspTensor),
                ("out", UspTensor), ("acc", UspTensor),
                ("lse", ctypes.c_void_p), ("lse_stride_b", ctypes.c_int64),
                ("lse_stride_h", ctypes.c_int64),
                ("merge_in", ctypes.c_int32), ("final_begin", ctypes.c_int32),
                ("final_end", ctypes.c_int32),
                ("seq_q", ctypes.c_void_p), ("seq_k", ctypes.c_void_p), ("sched", ctypes.c_void_p),
                ("flags", ctypes.c_int32), ("k_splits", ctypes.c_int32), ("workspace", ctypes.c_void_p),
                ("window_left", ctypes.c_int32), ("window_right", ctypes.c_int32)]


class UspBwdArgs(ctypes.Structure):
    _fields_ = [("dtype", ctypes.c_int32), ("B", ctypes.c_int32), ("Sq", ctypes.c_int32),
                ("Sk", ctypes.c_int32), ("Hq", ctypes.c_int32), ("Hkv", ctypes.c_int32),
                ("D", ctypes.c_int32), ("causal", ctypes.c_int32),
                ("softmax_scale", ctypes.c_float),
                ("dout", UspTensor), ("q", UspTensor), ("k", UspTensor), ("v", UspTensor),
                ("lse", ctypes.c_void_p), ("delta", ctypes.c_void_p),
                ("lse_stride_b", ctypes.c_int64), ("lse_stride_h", ctypes.c_int64),
                ("delta_stride_b", ctypes.c_int64), ("delta_stride_h", ctypes.c_int64),
                ("dq", UspTensor), ("dk", UspTensor), ("dv", UspTensor),
                ("accum_dq", ctypes.c_int32), ("accum_dk", ctypes.c_int32),
                ("accum_dv", ctypes.c_int32),
                ("dq16", UspTensor), ("dk16", UspTensor), ("dv16", UspTensor),
                ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_int64),
                ("seq_q", ctypes.c_void_p), ("seq_k", ctypes.c_void_p), ("total_k", ctypes.c_int64),
                ("sched", ctypes.c_void_p), ("flags", ctypes.c_int32),
                ("dq_splits", ctypes.c_int32), ("dkdv_splits", ctypes.c_int32),
                ("window_left", ctypes.c_int32), ("window_right", ctypes.c_int32),
                ("dkdv_heads", ctypes.c_int32)]


EXPORTS = ("usp_flash_fwd", "usp_flash_fwd_workspace_bytes", "usp_flash_bwd", "usp_flash_bwd_workspace_bytes", "usp_bwd_delta", "usp_lse_merge", "usp_copy_rows",
           "usp_cast_from_f32", "usp_add_f32", "usp_abi_version", "usp_strerror", "usp_last_launch_kinds", "usp_mfma_probe")


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load libusp_hip.so (once).  Raises RuntimeError, loudly, when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"libusp_hip.so not found at {_LIB_PATH}: the HIP extension is not built. Run "
            f"`python -c 'import __graft_entry__ as g; g.build()'` or "
            f"`make -C long-context-attention_amd/csrc`. There is no CPU fallback.")
    L = ctypes.CDLL(_LIB_PATH)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise RuntimeError(f"{_LIB_PATH} does not export {name} (stale build?)")
    L.usp_strerror.restype = ctypes.c_char_p
    L.usp_abi_version.restype = ctypes.c_int
    L.usp_last_launch_kinds.restype = ctypes.c_int
    L.usp_last_launch_kinds.argtypes = []
    if L.usp_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libusp_hip.so ABI {L.usp_abi_version()} != binding ABI {ABI_VERSION}")
    i32, i64, vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p
    L.usp_flash_fwd.argtypes = [ctypes.POINTER(UspFwdArgs), vp]
    L.usp_flash_bwd.argtypes = [ctypes.POINTER(UspBwdArgs), vp]
    L.usp_flash_fwd_workspace_bytes.argtypes = [ctypes.POINTER(UspFwdArgs), i32]
    L.usp_flash_fwd_workspace_bytes.restype = ctypes.c_int64
    L.usp_flash_bwd_workspace_bytes.argtypes = [ctypes.POINTER(UspBwdArgs)]
    L.usp_flash_bwd_workspace_bytes.restype = ctypes.c_int64
    L.usp_bwd_delta.argtypes = [i32, i32, i32, i32, i32, ctypes.POINTER(UspTensor),
                                ctypes.POINTER(UspTensor), vp, i64, i64, vp]
    L.usp_lse_merge.argtypes = [i32, i32, i32, i32, i32, ctypes.POINTER(UspTensor), vp, i64, i64,
                                ctypes.POINTER(UspTensor), vp, i64, i64, i32, vp]
    L.usp_copy_rows.argtypes = [vp, vp] + [i64] * 13 + [vp]
    L.usp_cast_from_f32.argtypes = [i32, vp, i64, vp, i64, i64, i64, vp]
    L.usp_add_f32.argtypes = [vp, i64, vp, i64, vp, i64, i64, i64, vp]
    L.usp_mfma_probe.argtypes = [vp, i64, i32, i32, vp, vp, vp]
    L.usp_mfma_probe.restype = ctypes.c_int
    for name in ("usp_flash_fwd", "usp_flash_bwd", "usp_bwd_delta", "usp_lse_merge", "usp_copy_rows",
                 "usp_cast_from_f32", "usp_add_f32"):
        getattr(L, name).restype = ctypes.c_int
    _lib = L
    return L


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {load().usp_strerror(rc).decode()} (code {rc})")


def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.bfloat16:
        return USP_BF16
    if dtype == torch.float16:
        return USP_FP16
    raise TypeError(f"libusp_hip supports bfloat16 / float16 inputs, got {dtype}")


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "libusp_hip kernels need ROCm device tensors (got a CPU tensor); there is no CPU "
                "fallback in this package")


def _t4(t: Optional[torch.Tensor]) -> UspTensor:
    """(B,S,H,D) view with unit dim stride -> usp_tensor."""
    if t is None:
        return UspTensor(None, 0, 0, 0)
    assert t.dim() == 4, t.shape
    if t.stride(3) != 1:
        raise ValueError("last (head_dim) stride must be 1")
    return UspTensor(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2))


def _lse3(t: torch.Tensor):
    """(B,H,S) fp32 with unit seq stride -> (ptr, stride_b, stride_h)."""
    assert t.dim() == 3 and t.dtype == torch.float32
    if t.shape[2] > 1 and t.stride(2) != 1:
        raise ValueError("lse/delta must have unit stride along the sequence")
    return ctypes.c_void_p(t.data_ptr()), t.stride(0), t.stride(1)


_FWD_WS = {}


def _parse_ksplit(raw) -> object:
    """USP_FWD_KSPLIT -> "auto" | 0 | n in 2..8.  Anything else counts as off, with one warning (a typo in an
    environment variable must not raise on every launch)."""
    raw = (raw or "auto").strip().lower()
    if raw in ("auto", "0"):
        return "auto" if raw == "auto" else 0
    try:
        n = int(raw)
    except ValueError:
        import warnings
        warnings.warn(f"USP_FWD_KSPLIT={raw!r} is neither 'auto', 0 nor an integer: the K split stays off")
        return 0
    return 0 if n < 2 else min(n, 8)


# Read ONCE, at import (a launch does not touch os.environ); set_fwd_ksplit changes it in-process (tests, A/B runs).
_KSPLIT_MODE = _parse_ksplit(os.environ.get("USP_FWD_KSPLIT"))


def set_fwd_ksplit(mode) -> object:
    """Set the K-split policy of dense forward launches ("auto" | 0 | n); returns the previous one."""
    global _KSPLIT_MODE
    prev, _KSPLIT_MODE = _KSPLIT_MODE, _parse_ksplit(str(mode))
    return prev


def fwd_k_splits(B: int, Sq: int, Hq: int, causal: bool) -> int:
    """How many work items a dense forward launch cuts every query tile's keys into (usp_fwd_args.k_splits).  A causal
    launch with fewer than two 256-row items per CU lasts as long as its heaviest item; cutting every item's keys into n
    equal runs (partials + one HBM-bound merge launch) fills the part.  Measured natively on MI355X (`kbench ksplit`,
    profiles/r02_kbench_ksplit*.log: B1 S16384 D128 causal, 2 heads 557 -> 936 TFLOP/s at n = 4, 4 heads 798 -> 1070 at
    n = 2, merge launch included; 8 heads fill the part and gain nothing) and, through this binding, by
    tests/test_gpu_parity.py::test_forward_k_split_through_the_binding.  Policy (USP_FWD_KSPLIT, read at import):
        auto (default)  causal launches of >= 4096 rows with fewer than two 256-row items per CU: n = 2 from 256 items
                        up, 4 below (the measured optima);
        n               n cuts (2..8) for every causal launch with fewer than two 256-row items per CU;
        0               off."""
    mode = _KSPLIT_MODE
    if mode == 0 or not causal:
        return 0
    items = B * Hq * ((Sq + 255) // 256)
    if items >= 512:
        return 0
    if mode == "auto":
        return 0 if Sq < 4096 else (2 if items >= 256 else 4)
    return mode


_BWD_SPLIT_MODE = _parse_ksplit(os.environ.get("USP_BWD_SPLIT"))      # same grammar as USP_FWD_KSPLIT


def set_bwd_split(mode) -> object:
    """Set the cut policy of dense backward launches ("auto" | 0 | n); returns the previous one."""
    global _BWD_SPLIT_MODE
    prev, _BWD_SPLIT_MODE = _BWD_SPLIT_MODE, _parse_ksplit(str(mode))
    return prev


def bwd_splits(B: int, Sq: int, Sk: int, Hq: int, causal: bool):
    """(dq_splits, dkdv_splits) of a dense backward call (usp_bwd_args, ABI v5): the dQ launch has B*Hq*ceil(Sq/256) work
    items and the dK/dV launch B*Hq*ceil(Sk/128) (one per query head and key block); with fewer than two per CU a
    causal launch lasts as long as its heaviest item, with fewer than one per CU any launch leaves CUs idle -- such items
    are cut (dQ along the keys, dK/dV along the query rows; partials + one reduce launch).  Measured on MI355X
    (`kbench bwd` with USP_KBENCH_BWD_SPLITS, profiles/r03_kbench_bwd_cuts.log), B1 S16384 D128 causal: 2 query heads
    394 -> 737 TFLOP/s at (4, 2), 4 heads 688 -> 808 at (2, 1); 8 heads (827) and 4 heads at S32768 (847) fill the part
    and lose 3-6 % to any cut.  Policy (USP_BWD_SPLIT = auto | 0 | n, read at import): auto = per launch, n = 2 from
    256 items up (causal only), 4 below; rows >= 4096 only."""
    mode = _BWD_SPLIT_MODE
    if mode == 0:
        return 0, 0

    def cuts(items, rows):
        if rows < 4096 or items >= 512 or (items >= 256 and not causal):
            return 0
        if mode != "auto":
            return mode
        return 2 if items >= 256 else 4
    return cuts(B * Hq * ((Sq + 255) // 256), Sq), cuts(B * Hq * ((Sk + 127) // 128), Sk)


_TLS = threading.local()      # per-thread cache of filled argument blocks (the autograd thread launches too)
_ARGS_CACHE_MAX = 64


def _strides(t):
    return None if t is None else t.stride()


def _window(window):
    """flash-attn's `window_size` -> (left, right) ints, or None for no window ((-1, -1) / None)."""
    if window is None:
        return None
    wl, wr = int(window[0]), int(window[1])
    return None if (wl < 0 and wr < 0) else (wl, wr)


def _fwd_args(q, k, v, softmax_scale, causal, lse, out, acc, merge_in, final_begin, final_end, interleave, n, window=None,
              family_flag=0):
    """The filled usp_fwd_args block of a dense launch.  Everything but the seven pointers is a function of
    (dtype, shapes, strides, flags), and a training loop issues the same few launches over and over: the block is
    cached per thread under that signature and only the pointers are patched (filling 27 ctypes fields and five
    usp_tensor structs costs ~35 us of host time per launch, tools/host_step_cpu.py; a hit costs ~8)."""
    key = (q.dtype, q.shape, q.stride(), k.shape, k.stride(), v.stride(), lse.stride(), _strides(out), _strides(acc),
           softmax_scale, causal, merge_in, final_begin, final_end, interleave, n, window, family_flag)
    cache = _TLS.__dict__.setdefault("fwd", {})
    a = cache.get(key)
    if a is None:
        B, Sq, Hq, D = q.shape
        a = UspFwdArgs()
        a.dtype = dtype_code(q.dtype)
        a.B, a.Sq, a.Sk, a.Hq, a.Hkv, a.D = B, Sq, k.shape[1], Hq, k.shape[2], D
        a.causal = 1 if causal else 0
        a.softmax_scale = float(softmax_scale)
        a.q, a.k, a.v, a.out, a.acc = _t4(q), _t4(k), _t4(v), _t4(out), _t4(acc)
        a.lse, a.lse_stride_b, a.lse_stride_h = _lse3(lse)
        a.merge_in = 1 if merge_in else 0
        a.final_begin = final_begin
        a.final_end = Sq if final_end is None else final_end
        a.flags = (USP_LAUNCH_INTERLEAVE if interleave else 0) | family_flag
        if window is not None:
            a.flags |= USP_ATTN_WINDOW
            a.window_left, a.window_right = window
        a.k_splits = n if n > 1 else 0
        if len(cache) >= _ARGS_CACHE_MAX:
            cache.clear()
        cache[key] = a
        return a
    a.q.ptr, a.k.ptr, a.v.ptr, a.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), lse.data_ptr()
    a.out.ptr = None if out is None else out.data_ptr()
    a.acc.ptr = None if acc is None else acc.data_ptr()
    return a


def flash_fwd(q, k, v, softmax_scale: float, causal: bool, lse, out=None, acc=None,
              merge_in: bool = False, final_begin: int = 0, final_end: Optional[int] = None,
              interleave: bool = False, k_splits: Optional[int] = None, window=None, family=None):
    """usp_flash_fwd (include/usp_hip.h).  q (B,Sq,Hq,D); k,v (B,Sk,Hkv,D); lse (B,Hq,Sq) fp32;
    out 16-bit / acc fp32 (B,Sq,Hq,D).  All may be strided views (unit dim stride).  `k_splits`: cut the keys of
    every query tile into that many work items (None: fwd_k_splits decides; 0 / 1: off).  `window` = flash-attn's
    window_size (left, right), None / (-1, -1) = none.  `family`: "row64" | "wave32" pins the kernel family of this call
    (ABI v6; None: set_kernel_family's default, normally "auto"); both families serve a K split."""
    _require_cuda(q, k, v, lse, out, acc)
    B, Sq, Hq, D = q.shape
    ff = _family_flag(family)
    n = fwd_k_splits(B, Sq, Hq, causal) if k_splits is None else int(k_splits)
    a = _fwd_args(q, k, v, softmax_scale, bool(causal), lse, out, acc, bool(merge_in), final_begin, final_end,
                  bool(interleave), n, _window(window), ff)
    L = load()
    if n > 1:
        # scratch for the partial results: one buffer per (device, stream), grown on demand; launches on one stream
        # are ordered, so the buffer is free again when the next launch on that stream starts
        need = L.usp_flash_fwd_workspace_bytes(ctypes.byref(a), n)
        key = (q.device.index, torch.cuda.current_stream().cuda_stream)
        ws = _FWD_WS.get(key)
        if ws is None or ws.numel() < need:
            ws = _FWD_WS[key] = torch.empty(need, dtype=torch.uint8, device=q.device)
        a.workspace = ws.data_ptr()
    _check(L.usp_flash_fwd(ctypes.byref(a), _stream()), "usp_flash_fwd")


def _t3(t: Optional[torch.Tensor]) -> UspTensor:
    """(T,H,D) token tensor with unit dim stride -> usp_tensor (batch stride unused)."""
    if t is None:
        return UspTensor(None, 0, 0, 0)
    assert t.dim() == 3, t.shape
    if t.stride(2) != 1:
        raise ValueError("last (head_dim) stride must be 1")
    return UspTensor(t.data_ptr(), 0, t.stride(0), t.stride(1))


def _lse2(t: torch.Tensor):
    """(H,T) fp32 with unit token stride -> (ptr, 0, stride_h)."""
    assert t.dim() == 2 and t.dtype == torch.float32
    if t.shape[1] > 1 and t.stride(1) != 1:
        raise ValueError("lse/delta must have unit stride along the tokens")
    return ctypes.c_void_p(t.data_ptr()), 0, t.stride(0)


def _seq(t: torch.Tensor, n: int) -> ctypes.c_void_p:
    """(n,2) int32 device tensor of (first_row, rows) pairs."""
    if t.dtype != torch.int32 or tuple(t.shape) != (n, 2) or not t.is_contiguous():
        raise ValueError("sequence table must be a contiguous (num_seq, 2) int32 tensor")
    _require_cuda(t)
    return ctypes.c_void_p(t.data_ptr())


_SCHED = {}


def sched_block(device) -> torch.Tensor:
    """The 16-int32 control block of the packed kernels' dynamic item queue (include/usp_hip.h `sched`):
    zeroed once, left zeroed by every launch, one per (device, stream) because launches on different
    streams may overlap."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    blk = _SCHED.get(key)
    if blk is None:
        blk = _SCHED[key] = torch.zeros(16, dtype=torch.int32, device=device)
    return blk


def flash_fwd_packed(q, k, v, seq_q, seq_k, max_q: int, max_k: int, softmax_scale: float,
                     causal: bool, lse, out=None, acc=None, merge_in: bool = False,
                     final_begin: int = 0, final_end: int = 2, interleave: bool = False):
    """usp_flash_fwd in packed variable-length mode.  q/out/acc (T,Hq,D), k/v (T',Hkv,D), lse (Hq,T)
    fp32; seq_q/seq_k (num_seq,2) int32 device tables of (first_row, rows); final_begin/final_end
    count half sequences (0,1,2)."""
    _require_cuda(q, k, v, lse, out, acc)
    n = seq_q.shape[0]
    a = UspFwdArgs()
    a.dtype = dtype_code(q.dtype)
    a.B, a.Sq, a.Sk, a.Hq, a.Hkv, a.D = n, int(max_q), int(max_k), q.shape[1], k.shape[1], q.shape[2]
    a.causal = 1 if causal else 0
    a.softmax_scale = float(softmax_scale)
    a.q, a.k, a.v, a.out, a.acc = _t3(q), _t3(k), _t3(v), _t3(out), _t3(acc)
    a.lse, a.lse_stride_b, a.lse_stride_h = _lse2(lse)
    a.merge_in = 1 if merge_in else 0
    a.final_begin, a.final_end = int(final_begin), int(final_end)
    a.seq_q, a.seq_k = _seq(seq_q, n), _seq(seq_k, n)
    a.sched = sched_block(q.device).data_ptr()
    a.flags = USP_LAUNCH_INTERLEAVE if interleave else 0
    _check(load().usp_flash_fwd(ctypes.byref(a), _stream()), "usp_flash_fwd")


def flash_bwd_packed(dout, q, k, v, lse, delta, seq_q, seq_k, max_q: int, max_k: int, dq, dk, dv,
                     softmax_scale: float, causal: bool, accum_dq=False, accum_dk=False,
                     accum_dv=False, dq16=None, dk16=None, dv16=None, interleave: bool = False):
    """usp_flash_bwd in packed variable-length mode (layouts as flash_fwd_packed; lse/delta (Hq,T))."""
    _require_cuda(dout, q, k, v, lse, delta, dq, dk, dv, dq16, dk16, dv16)
    n = seq_q.shape[0]
    a = UspBwdArgs()
    a.dtype = dtype_code(q.dtype)
    a.B, a.Sq, a.Sk, a.Hq, a.Hkv, a.D = n, int(max_q), int(max_k), q.shape[1], k.shape[1], q.shape[2]
    a.causal = 1 if causal else 0
    a.softmax_scale = float(softmax_scale)
    a.dout, a.q, a.k, a.v = _t3(dout), _t3(q), _t3(k), _t3(v)
    a.lse, a.lse_stride_b, a.lse_stride_h = _lse2(lse)
    a.delta, a.delta_stride_b, a.delta_stride_h = _lse2(delta)
    for t in (dq, dk, dv):
        if t is not None and t.dtype != torch.float32:
            raise TypeError("dq/dk/dv buffers of usp_flash_bwd are fp32")
    a.dq, a.dk, a.dv = _t3(dq), _t3(dk), _t3(dv)
    a.dq16, a.dk16, a.dv16 = _t3(dq16), _t3(dk16), _t3(dv16)
    a.accum_dq, a.accum_dk, a.accum_dv = int(bool(accum_dq)), int(bool(accum_dk)), int(bool(accum_dv))
    a.seq_q, a.seq_k = _seq(seq_q, n), _seq(seq_k, n)
    a.sched = sched_block(q.device).data_ptr()
    a.flags = USP_LAUNCH_INTERLEAVE if interleave else 0
    a.total_k = k.shape[0]
    L = load()
    need = L.usp_flash_bwd_workspace_bytes(ctypes.byref(a))
    ws = None
    if need > 0:
        ws = torch.empty(need, dtype=torch.uint8, device=q.device)
        a.workspace, a.workspace_bytes = ws.data_ptr(), need
    _check(L.usp_flash_bwd(ctypes.byref(a), _stream()), "usp_flash_bwd")


def bwd_delta(dout, out, delta):
    _require_cuda(dout, out, delta)
    B, S, H, D = dout.shape
    td, to = _t4(dout), _t4(out)
    p, sb, sh = _lse3(delta)
    _check(load().usp_bwd_delta(dtype_code(dout.dtype), B, S, H, D, ctypes.byref(td),
                                ctypes.byref(to), p, sb, sh, _stream()), "usp_bwd_delta")


def flash_bwd(dout, q, k, v, lse, delta, dq, dk, dv, softmax_scale: float, causal: bool,
              accum_dq=False, accum_dk=False, accum_dv=False, dq16=None, dk16=None, dv16=None,
              interleave: bool = False, splits=None, window=None, family=None, only=None, dkdv_heads: int = 0):
    """usp_flash_bwd.  dq/dk/dv are fp32 (B,S,H,D) views, written or accumulated; a 16-bit
    dq16/dk16/dv16 receives the FINAL rounded result instead (the fp32 tensor may then be None
    unless it is accumulated from).  `splits` = (dq_splits, dkdv_splits), None: bwd_splits decides.  `window` =
    flash-attn's window_size (left, right), None / (-1, -1) = none.  `family`: as flash_fwd.  `only`: "dkdv" | "dq" issues
    just that launch of the two (ABI v6: USP_BWD_SKIP_DQ / USP_BWD_SKIP_DKDV).  `dkdv_heads` (ABI v7): query heads of a KV
    group one dK/dV work item streams (a divisor of Hq / Hkv; 0 = the library decides)."""
    _require_cuda(dout, q, k, v, lse, delta, dq, dk, dv, dq16, dk16, dv16)
    B, Sq, Hq, D = q.shape
    Sk, Hkv = k.shape[1], k.shape[2]
    a = UspBwdArgs()
    a.dtype = dtype_code(q.dtype)
    a.B, a.Sq, a.Sk, a.Hq, a.Hkv, a.D = B, Sq, Sk, Hq, Hkv, D
    a.causal = 1 if causal else 0
    a.softmax_scale = float(softmax_scale)
    a.dout, a.q, a.k, a.v = _t4(dout), _t4(q), _t4(k), _t4(v)
    a.lse, a.lse_stride_b, a.lse_stride_h = _lse3(lse)
    a.delta, a.delta_stride_b, a.delta_stride_h = _lse3(delta)
    for t in (dq, dk, dv):
        if t is not None and t.dtype != torch.float32:
            raise TypeError("dq/dk/dv buffers of usp_flash_bwd are fp32")
    a.dq, a.dk, a.dv = _t4(dq), _t4(dk), _t4(dv)
    a.dq16, a.dk16, a.dv16 = _t4(dq16), _t4(dk16), _t4(dv16)
    a.accum_dq, a.accum_dk, a.accum_dv = int(bool(accum_dq)), int(bool(accum_dk)), int(bool(accum_dv))
    ff = _family_flag(family)
    a.flags = (USP_LAUNCH_INTERLEAVE if interleave else 0) | ff | {None: 0, "dkdv": USP_BWD_SKIP_DQ, "dq": USP_BWD_SKIP_DKDV}[only]
    a.dq_splits, a.dkdv_splits = bwd_splits(B, Sq, Sk, Hq, bool(causal)) if splits is None else splits
    a.dkdv_heads = int(dkdv_heads)
    win = _window(window)
    if win is not None:
        a.flags |= USP_ATTN_WINDOW
        a.window_left, a.window_right = win
    L = load()
    need = L.usp_flash_bwd_workspace_bytes(ctypes.byref(a))     # GQA head split and / or cuts of few-item launches
    ws = None
    if need > 0:
        ws = torch.empty(need, dtype=torch.uint8, device=q.device)   # caching allocator; stream-ordered
        a.workspace, a.workspace_bytes = ws.data_ptr(), need
    _check(L.usp_flash_bwd(ctypes.byref(a), _stream()), "usp_flash_bwd")


def lse_merge(acc, lse, blk_out, blk_lse, first: bool):
    _require_cuda(acc, lse, blk_out, blk_lse)
    B, S, H, D = acc.shape
    ta, tb = _t4(acc), _t4(blk_out)
    p, sb, sh = _lse3(lse)
    p2, sb2, sh2 = _lse3(blk_lse)
    _check(load().usp_lse_merge(dtype_code(blk_out.dtype), B, S, H, D, ctypes.byref(ta), p, sb, sh,
                                ctypes.byref(tb), p2, sb2, sh2, 1 if first else 0, _stream()),
           "usp_lse_merge")


def copy_rows(dst, src, row_bytes, sizes, dst_strides, src_strides):
    """usp_copy_rows: strides in BYTES, 4 outer dims."""
    _require_cuda(dst, src)
    n = list(sizes) + [1] * (4 - len(sizes))
    ds = list(dst_strides) + [0] * (4 - len(dst_strides))
    ss = list(src_strides) + [0] * (4 - len(src_strides))
    _check(load().usp_copy_rows(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()),
                                row_bytes, *n, *ds, *ss, _stream()), "usp_copy_rows")


def _rows2d(t: torch.Tensor):
    """View a tensor as `rows` rows of `n` contiguous elements with one row stride."""
    if t.is_contiguous():
        return 1, t.numel(), t.numel()
    if t.dim() == 4 and t[0].is_contiguous():          # (B, ...) slices of a bigger batch stride
        return t.shape[0], t[0].numel(), t.stride(0)
    raise ValueError("expected a contiguous tensor or a batch of contiguous slices")


def cast_from_f32(dst16, src32):
    _require_cuda(dst16, src32)
    r1, n1, s1 = _rows2d(dst16)
    r2, n2, s2 = _rows2d(src32)
    if (r1, n1) != (r2, n2):
        r1, n1, s1, s2 = dst16.shape[0], dst16[0].numel(), dst16.stride(0), src32.stride(0)
        assert dst16[0].is_contiguous() and src32[0].is_contiguous()
    _check(load().usp_cast_from_f32(dtype_code(dst16.dtype), ctypes.c_void_p(dst16.data_ptr()), s1,
                                    ctypes.c_void_p(src32.data_ptr()), s2, r1, n1, _stream()),
           "usp_cast_from_f32")


def add_f32(dst, a, b):
    """dst = a + b (fp32), tensors contiguous or batches of contiguous slices of equal shape."""
    _require_cuda(dst, a, b)
    assert dst.shape == a.shape == b.shape
    if dst.is_contiguous() and a.is_contiguous() and b.is_contiguous():
        rows, n, sd, sa, sb_ = 1, dst.numel(), dst.numel(), a.numel(), b.numel()
    else:
        assert dst[0].is_contiguous() and a[0].is_contiguous() and b[0].is_contiguous()
        rows, n = dst.shape[0], dst[0].numel()
        sd, sa, sb_ = dst.stride(0), a.stride(0), b.stride(0)
    _check(load().usp_add_f32(ctypes.c_void_p(dst.data_ptr()), sd, ctypes.c_void_p(a.data_ptr()), sa,
                              ctypes.c_void_p(b.data_ptr()), sb_, rows, n, _stream()), "usp_add_f32")


def mfma_probe(operands: torch.Tensor, iters: int, waves_per_simd: int = 1, clocks: Optional[torch.Tensor] = None):
    """usp_mfma_probe: one launch of the MFMA-only loop on `operands` (a bf16 device tensor of >= 32768 elements).
    Returns the FLOPs of the launch; `clocks`: optional int64[2] device tensor (shader-clock and 100 MHz ticks)."""
    _require_cuda(operands, clocks)
    assert operands.dtype == torch.bfloat16 and operands.is_contiguous() and operands.numel() >= 32768
    key = ("probe_sink", operands.device.index)
    sink = _SCHED.get(key)
    if sink is None:
        sink = _SCHED[key] = torch.zeros(512, dtype=torch.float32, device=operands.device)
    _check(load().usp_mfma_probe(ctypes.c_void_p(operands.data_ptr()), operands.numel() * 2, int(iters), int(waves_per_simd),
                                 ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(clocks.data_ptr() if clocks is not None else None),
                                 _stream()), "usp_mfma_probe")
    cus = torch.cuda.get_device_properties(operands.device).multi_processor_count
    return cus * 4 * waves_per_simd * iters * 64 * 32768.0
