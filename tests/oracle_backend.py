"""TEST-ONLY block backend: implements the BlockBackend seam of yunchang_amd.kernels.attention with
the CPU oracle (oracle/usp_oracle.py), so the distributed orchestration (all-to-all, ring relay,
schedule, autograd glue) can run on CPU tensors under gloo.  Never installed by the package."""
import numpy as np
import torch

from oracle import usp_oracle as O


def _np(t):
    return t.detach().to(torch.float64).numpy()


def _put(dst, arr):
    dst.copy_(torch.from_numpy(np.ascontiguousarray(arr)).to(dst.dtype))


def _operand(t, what, ptr_align=16, stride_align_bytes=16):
    """The operand constraints of the device kernels (csrc/usp_flash_fwd.hip:tensor16_ok, usp_flash_bwd.hip): unit
    head-dim stride, pointer and every other stride a multiple of 16 bytes (8 for 16-bit outputs).  Asserted here so
    that the CPU orchestration tests fail where the HIP path would return USP_EUNSUPPORTED."""
    if t is None:
        return
    es = t.element_size()
    assert t.stride(-1) == 1, f"{what}: head-dim stride {t.stride(-1)}"
    assert (t.storage_offset() * es) % ptr_align == 0, f"{what}: view starts {t.storage_offset() * es} bytes into its buffer"
    for d in range(t.dim() - 1):
        assert t.shape[d] == 1 or (t.stride(d) * es) % stride_align_bytes == 0, \
            f"{what}: stride {t.stride(d)} of dim {d} is not a multiple of {stride_align_bytes} bytes"


class OracleBlockBackend:
    name = "oracle"

    def __init__(self):
        self.calls = []

    def fwd(self, q, k, v, softmax_scale, causal, lse, out=None, acc=None, merge_in=False,
            final_begin=0, final_end=None, window=None, k_splits=None):
        Sq = q.shape[1]                 # (k_splits: how the device fills its CUs; the result does not depend on it)
        window = (-1, -1) if window is None else window
        fe = Sq if final_end is None else final_end
        for t, what in ((q, "q"), (k, "k"), (v, "v"), (acc, "acc")):
            _operand(t, what)
        _operand(out, "out", 8, 8)
        assert lse.stride(-1) == 1 or lse.shape[-1] == 1, "lse needs unit stride along the sequence"
        # what the C side checks before a launch (usp_flash_fwd): shapes, head dims, dtypes
        B, _, Hq, D = q.shape
        assert D in (32, 64, 128) and k.shape == v.shape and k.shape[0] == B and k.shape[3] == D and Hq % k.shape[2] == 0
        assert q.dtype in (torch.bfloat16, torch.float16) and k.dtype == q.dtype == v.dtype
        assert lse.dtype == torch.float32 and tuple(lse.shape) == (B, Hq, Sq)
        assert out is None or (out.dtype == q.dtype and out.shape == q.shape)
        assert acc is None or (acc.dtype == torch.float32 and acc.shape == q.shape)
        assert (fe <= final_begin or out is not None) and ((final_begin <= 0 and fe >= Sq) or acc is not None) \
            and (not merge_in or acc is not None), "final rows need `out`, the others (and a merge) need `acc`"
        self.calls.append(("fwd", tuple(q.shape), tuple(k.shape), bool(causal), bool(merge_in),
                           final_begin, fe))
        bo, bl = O.attention_ref(_np(q), _np(k), _np(v), causal, softmax_scale, window=window)      # (B,Sq,H,D), (B,H,Sq)
        if merge_in:
            o_run = _np(acc)
            l_run = np.swapaxes(_np(lse), 1, 2)[..., None]                       # (B,S,H,1)
            o_new, l_new = O.update_out_and_lse(o_run, l_run, bo, bl)
            bl = np.swapaxes(l_new[..., 0], 1, 2)
            bo = o_new
        _put(lse, bl)
        if fe > final_begin:
            _put(out[:, final_begin:fe], bo[:, final_begin:fe])
        if final_begin > 0:
            _put(acc[:, :final_begin], bo[:, :final_begin])
        if fe < Sq:
            _put(acc[:, fe:], bo[:, fe:])

    def delta(self, dout, out, delta):
        _operand(dout, "dout"); _operand(out, "out")
        assert delta.stride(-1) == 1 or delta.shape[-1] == 1
        _put(delta, np.einsum("bshd,bshd->bhs", _np(dout), _np(out)))

    def bwd(self, dout, q, k, v, lse, delta, dq, dk, dv, softmax_scale, causal, accum_dq=False,
            accum_dk=False, accum_dv=False, dq16=None, dk16=None, dv16=None, window=None, only=None):
        window = (-1, -1) if window is None else window
        assert only in (None, "dq", "dkdv")
        for t, what in ((dout, "dout"), (q, "q"), (k, "k"), (v, "v"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
            _operand(t, what)
        for t, what in ((dq16, "dq16"), (dk16, "dk16"), (dv16, "dv16")):
            _operand(t, what, 8, 8)
        assert q.shape[3] in (32, 64, 128) and q.dtype in (torch.bfloat16, torch.float16) and dout.dtype == q.dtype
        assert lse.dtype == torch.float32 and delta.dtype == torch.float32 and lse.shape == delta.shape
        wanted = {None: ("dq", "dk", "dv"), "dq": ("dq",), "dkdv": ("dk", "dv")}[only]    # (a skipped launch needs no outputs and touches none)
        for g32, g16, ref, nm in ((dq, dq16, q, "dq"), (dk, dk16, k, "dk"), (dv, dv16, v, "dv")):
            if nm not in wanted:
                continue
            assert g32 is not None or g16 is not None, f"{nm}: no destination"
            assert g32 is None or (g32.dtype == torch.float32 and g32.shape == ref.shape), nm
            assert g16 is None or (g16.dtype == q.dtype and g16.shape == ref.shape), nm
        self.calls.append(("bwd", tuple(q.shape), tuple(k.shape), bool(causal)) + ((only,) if only else ()))
        # block_bwd derives delta from `out`; feed it an `out` whose rowsum(dout*out) equals the
        # supplied delta is not possible in general, so restate with delta directly:
        qn, kn, vn, don = _np(q), _np(k), _np(v), _np(dout)
        ln, dl = _np(lse), _np(delta)
        B, Sq, Hq, D = qn.shape
        Sk, Hkv = kn.shape[1], kn.shape[2]
        g = Hq // Hkv
        s = O._scores(qn, kn, softmax_scale, causal, window)
        fin = np.isfinite(ln)
        p = np.where(fin[..., None], np.exp(s - np.where(fin, ln, 0.0)[..., None]), 0.0)
        vv, kk = np.repeat(vn, g, axis=2), np.repeat(kn, g, axis=2)
        gdv = np.einsum("bhts,bthd->bshd", p, don).reshape(B, Sk, Hkv, g, D).sum(3)
        dp = np.einsum("bthd,bshd->bhts", don, vv)
        ds = p * (dp - dl[..., None]) * softmax_scale
        gdq = np.einsum("bhts,bshd->bthd", ds, kk)
        gdk = np.einsum("bhts,bthd->bshd", ds, qn).reshape(B, Sk, Hkv, g, D).sum(3)
        for nm, dst, d16, val, accum in (("dq", dq, dq16, gdq, accum_dq), ("dk", dk, dk16, gdk, accum_dk),
                                         ("dv", dv, dv16, gdv, accum_dv)):
            if nm not in wanted:
                continue
            tot = val + _np(dst) if accum else val
            _put(d16 if d16 is not None else dst, tot)

    # packed variable-length mode (include/usp_hip.h): every sequence is one B = 1 dense call on views
    def fwd_packed(self, q, k, v, seq_q, seq_k, max_q, max_k, softmax_scale, causal, lse, out=None,
                   acc=None, merge_in=False, final_begin=0, final_end=2):
        for (qf, ql), (kf, kl) in zip(seq_q.tolist(), seq_k.tolist()):
            assert ql <= max_q and kl <= max_k
            if ql <= 0:
                continue
            half = ql // 2
            fb = ql if final_begin >= 2 else final_begin * half
            fe = ql if final_end >= 2 else final_end * half
            self.fwd(q[None, qf:qf + ql], k[None, kf:kf + kl], v[None, kf:kf + kl], softmax_scale, causal,
                     lse[None, :, qf:qf + ql], None if out is None else out[None, qf:qf + ql],
                     None if acc is None else acc[None, qf:qf + ql], merge_in, fb, fe)

    def bwd_packed(self, dout, q, k, v, lse, delta, seq_q, seq_k, max_q, max_k, dq, dk, dv,
                   softmax_scale, causal, accum_dq=False, accum_dk=False, accum_dv=False, dq16=None,
                   dk16=None, dv16=None):
        cut = lambda t, a, n: None if t is None else t[None, a:a + n]
        for (qf, ql), (kf, kl) in zip(seq_q.tolist(), seq_k.tolist()):
            assert ql <= max_q and kl <= max_k
            if ql <= 0 or kl <= 0:
                continue
            self.bwd(cut(dout, qf, ql), cut(q, qf, ql), cut(k, kf, kl), cut(v, kf, kl),
                     lse[None, :, qf:qf + ql], delta[None, :, qf:qf + ql], cut(dq, qf, ql), cut(dk, kf, kl),
                     cut(dv, kf, kl), softmax_scale, causal, accum_dq, accum_dk, accum_dv,
                     cut(dq16, qf, ql), cut(dk16, kf, kl), cut(dv16, kf, kl))

    def merge(self, acc, lse, blk_out, blk_lse, first):
        if first:
            _put(acc, _np(blk_out)); _put(lse, _np(blk_lse))
            return
        l_run = np.swapaxes(_np(lse), 1, 2)[..., None]
        o_new, l_new = O.update_out_and_lse(_np(acc), l_run, _np(blk_out), _np(blk_lse))
        _put(acc, o_new); _put(lse, np.swapaxes(l_new[..., 0], 1, 2))

    @staticmethod
    def _rows(t):
        """The layout constraint of the device kernels (yunchang_amd/_C.py:_rows2d): contiguous, or a batch of
        contiguous slices.  Asserted here too, so the CPU orchestration tests fail where the HIP path would."""
        assert t.is_contiguous() or (t.dim() == 4 and t[0].is_contiguous()), \
            f"usp_cast_from_f32 / usp_add_f32 need contiguous rows, got shape {tuple(t.shape)} strides {t.stride()}"

    def cast(self, dst16, src32):
        self._rows(dst16); self._rows(src32)
        dst16.copy_(src32.to(dst16.dtype))

    def add(self, dst, a, b):
        for t in (dst, a, b):
            self._rows(t)
        torch.add(a, b, out=dst)

    def copy_rows(self, dst, src, row_bytes, sizes, dst_strides, src_strides):
        raise AssertionError("host tensors take the as_strided path in comm/all_to_all.py")
