#!/usr/bin/env python3
"""Generate golden fixtures by running the REFERENCE ITSELF (yunchang, imported read-only from
/root/reference) on CPU under torch.distributed/gloo.

Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Recipe = SURVEY.md Appendix A.  The reference's orchestration (set_seq_parallel_pg,
EXTRACT_FUNC_DICT, LongContextAttention, SeqAllToAll4D, zigzag/basic ring, RingComm,
update_out_and_lse) runs UNMODIFIED.  Two third-party seams are substituted because their
kernels do not exist on CPU:
  (1) aten::_scaled_dot_product_efficient_attention (kernels/attention.py:6,76-86; CUDA only)
      -> aten::_scaled_dot_product_flash_attention_for_cpu (same (B,H,S,D) views, out + fp32 lse)
  (2) pytorch_attn_backward (kernels/attention.py:138-159; raises) -> an fp32 torch block
      backward with the flash_attn_backward argument contract (kernels/attention.py:205-206).
Backward fixtures are 16-bit only: the reference's fp32 accumulators alias for fp32 inputs
(zigzag_ring_flash_attn.py:147-149).

  (3) transport: gloo cannot send to self, NCCL/RCCL can.  At ring degree 1 the reference's
      backward still posts a self send/recv of dk/dv (ring_flash_attn.py:141-143); for that case
      only, RingComm.commit is replaced by the equivalent local copy.

Inputs are N(0,1) like the reference protocol (test/test_hybrid_attn.py:125-184) but drawn from
``np.random.RandomState(seed)`` (legacy generator: bit-stable across numpy versions) so that
fixtures only need to store the topology, the seed and every rank's local out / dq / dk / dv
exactly as the reference returned them (16-bit tensors as raw bit patterns).
"""
import os
import sys

sys.dont_write_bytecode = True
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
os.environ.setdefault("OMP_NUM_THREADS", "1")

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

# name, ws, ud, rd, impl, B, S, Hq, Hkv, D, dtype, bwd
CASES = [
    # C1 shape (BASELINE.json configs[0]): single process, causal, forward
    ("c1_w1_fp32", 1, 1, 1, "basic", 1, 1024, 8, 8, 64, "float32", False),
    ("c1_w1_bf16", 1, 1, 1, "basic", 1, 1024, 8, 8, 64, "bfloat16", False),
    # C2 topology (ring=1, ulysses=1), batch 2
    ("c2_w1_bf16", 1, 1, 1, "basic", 2, 256, 2, 2, 128, "bfloat16", True),
    # C3 topology: ulysses=2, ring=1
    ("c3_w2_u2r1_bf16", 2, 2, 1, "basic", 1, 256, 4, 4, 128, "bfloat16", True),
    # C4 topology: ulysses=1, ring=4 zigzag
    ("c4_w4_u1r4_bf16", 4, 1, 4, "zigzag", 1, 512, 2, 2, 128, "bfloat16", True),
    # basic (contiguous) ring with ring=2 for the non-zigzag schedule
    ("b_w2_u1r2_bf16", 2, 1, 2, "basic", 1, 512, 2, 2, 64, "bfloat16", True),
    # C5 topology: ulysses=2 x ring=4 zigzag, GQA
    ("c5_w8_u2r4_gqa_bf16", 8, 2, 4, "zigzag", 1, 512, 4, 2, 128, "bfloat16", True),
    ("c5_w8_u2r4_gqa_fp16", 8, 2, 4, "zigzag", 1, 512, 4, 2, 64, "float16", True),
    # SURVEY 8(f) "next" rows: stripe ring, UlyssesAttention, packed-qkv layer (layer suffix after ':')
    ("n_w4_u1r4_strip_bf16", 4, 1, 4, "strip", 1, 512, 2, 2, 64, "bfloat16", True),
    ("n_w4_u2r2_strip_bf16", 4, 2, 2, "strip", 1, 256, 4, 4, 64, "bfloat16", True),
    ("n_w2_ulysses_bf16:ulysses", 2, 2, 1, "basic", 2, 256, 4, 2, 64, "bfloat16", True),
    ("n_w4_u2r2_qkvpacked_bf16:qkvpacked", 4, 2, 2, "zigzag", 1, 256, 4, 4, 64, "bfloat16", True),
    # NON-causal basic ring with ring degree > 1 (every step computes: ring_flash_attn.py:35 `if not causal or
    # step <= comm.rank`; an optional 13th field = causal), the second one with batch 2 and GQA on a
    # ulysses x ring grid (seq-major views with a real batch stride reach the ring)
    ("f_w4_u1r4_full_bf16", 4, 1, 4, "basic", 1, 512, 2, 2, 64, "bfloat16", True, False),
    ("f_w4_u2r2_full_b2_gqa_bf16", 4, 2, 2, "basic", 2, 256, 4, 2, 64, "bfloat16", True, False),
]
SEED = 0


def make_inputs(B, S, Hq, Hkv, D, seed=SEED):
    """Global q,k,v,dout as float32 numpy N(0,1); tests regenerate them with this function."""
    rs = np.random.RandomState(seed)
    q = rs.standard_normal((B, S, Hq, D)).astype(np.float32)
    k = rs.standard_normal((B, S, Hkv, D)).astype(np.float32)
    v = rs.standard_normal((B, S, Hkv, D)).astype(np.float32)
    dout = rs.standard_normal((B, S, Hq, D)).astype(np.float32)
    return q, k, v, dout


def _block_bwd_torch(dout, q, k, v, out, softmax_lse, dq_buf, dk_buf, dv_buf, dropout_p,
                     softmax_scale, bwd_causal, window_size, softcap, alibi_slopes,
                     deterministic, rng_state=None, *a, **kw):
    """fp32 block backward written against the contract of kernels/attention.py:205-206."""
    B, Sq, Hq, D = q.shape
    Sk, Hkv = k.shape[1], k.shape[2]
    g = Hq // Hkv
    scale = D ** -0.5 if softmax_scale is None else softmax_scale
    qf, kf, vf, of, dof = (t.float() for t in (q, k, v, out, dout))
    kk = kf.repeat_interleave(g, dim=2)
    vv = vf.repeat_interleave(g, dim=2)
    s = torch.einsum("bthd,bshd->bhts", qf, kk) * scale
    if bwd_causal:
        row = torch.arange(Sq)[:, None]
        col = torch.arange(Sk)[None, :]
        s = s.masked_fill(col > row + Sk - Sq, float("-inf"))
    p = torch.exp(s - softmax_lse.float()[..., None])
    dv = torch.einsum("bhts,bthd->bshd", p, dof)
    dp = torch.einsum("bthd,bshd->bhts", dof, vv)
    delta = (dof * of).sum(-1).permute(0, 2, 1)
    ds = p * (dp - delta[..., None]) * scale
    dq = torch.einsum("bhts,bshd->bthd", ds, kk)
    dk = torch.einsum("bhts,bthd->bshd", ds, qf)
    dq_buf.copy_(dq.to(dq_buf.dtype))
    dk_buf.copy_(dk.reshape(B, Sk, Hkv, g, D).sum(3).to(dk_buf.dtype))
    dv_buf.copy_(dv.reshape(B, Sk, Hkv, g, D).sum(3).to(dv_buf.dtype))


def _patch_reference():
    sys.path.insert(0, REF)
    import yunchang.kernels.attention as A
    import yunchang.kernels as K

    def eff(q, k, v, attn_bias=None, compute_log_sumexp=True, dropout_p=0.0, is_causal=False,
            scale=None):
        return torch.ops.aten._scaled_dot_product_flash_attention_for_cpu(
            q, k, v, dropout_p, is_causal, scale=scale)[:2]

    A._scaled_dot_product_efficient_attention = eff
    K.pytorch_attn_backward = _block_bwd_torch

    import yunchang.ring.utils as U
    orig_commit = U.RingComm.commit

    def commit(self):
        if self.world_size != 1:
            return orig_commit(self)
        if self._reqs is not None:
            raise RuntimeError("commit called twice")
        ops = self._ops                                  # [isend(x), irecv(y), isend(x2), irecv(y2), ...]
        for snd, rcv in zip(ops[0::2], ops[1::2]):
            rcv.tensor.copy_(snd.tensor)
        self._reqs = []

    U.RingComm.commit = commit


def _bits(t: torch.Tensor) -> np.ndarray:
    if t.dtype == torch.float32:
        return t.detach().contiguous().numpy()
    return t.detach().contiguous().view(torch.int16).numpy().view(np.uint16)


def _worker(rank, ws, case, port, ret):
    name, _, ud, rd, impl, B, S, Hq, Hkv, D, dtype_s, bwd = case[:12]
    causal = case[12] if len(case) > 12 else True
    layer = name.split(":")[1] if ":" in name else "hybrid"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    torch.set_num_threads(1)
    _patch_reference()
    from yunchang import LongContextAttention, set_seq_parallel_pg, EXTRACT_FUNC_DICT
    from yunchang.kernels import AttnType

    dtype = getattr(torch, dtype_s)
    q, k, v, dout = (torch.from_numpy(t).to(dtype) for t in make_inputs(B, S, Hq, Hkv, D))
    for t in (q, k, v, dout):                         # test/test_hybrid_attn.py:181-184
        dist.broadcast(t, src=0)

    set_seq_parallel_pg(ud, rd, rank, ws)
    ext = EXTRACT_FUNC_DICT[impl]
    lq, lk, lv, ldo = (ext(t, rank, world_size=ws, rd=rd, ud=ud).detach().clone()
                       for t in (q, k, v, dout))
    if bwd:
        lq.requires_grad_(True); lk.requires_grad_(True); lv.requires_grad_(True)
    kw = dict(dropout_p=0, causal=causal, window_size=(-1, -1), softcap=0.0, alibi_slopes=None,
              deterministic=False, return_attn_probs=True)
    if layer == "hybrid":
        attn = LongContextAttention(ring_impl_type=impl, attn_type=AttnType.TORCH_EFFICIENT)
        out = attn(lq, lk, lv, **kw)
    elif layer == "ulysses":
        # the reference's fwd-bwd stage for TORCH_* is its picotron-style ring_pytorch_attn_func,
        # whose backward raises (ring_pytorch_attn.py:103); UlyssesAttention's semantics are exactly
        # "all-to-all -> one local attention -> all-to-all", so the local attention is supplied as an
        # autograd-aware torch function (fp32 attention_ref maths) through the same selector seam.
        import yunchang.ulysses.attn_layer as UA
        from yunchang.globals import PROCESS_GROUP

        def local_attn(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, **_):
            g = q.shape[2] // k.shape[2]
            kk, vv = k.float().repeat_interleave(g, 2), v.float().repeat_interleave(g, 2)
            s = torch.einsum("bthd,bshd->bhts", q.float() * softmax_scale, kk)
            if causal:
                Sq, Sk = q.shape[1], k.shape[1]
                s = s.masked_fill(torch.arange(Sk)[None, :] > torch.arange(Sq)[:, None] + Sk - Sq, float("-inf"))
            return torch.einsum("bhts,bshd->bthd", s.softmax(-1), vv).to(q.dtype)
        UA.select_flash_attn_impl = lambda *a, **k: local_attn
        attn = UA.UlyssesAttention(PROCESS_GROUP.ULYSSES_PG, attn_type=AttnType.TORCH_EFFICIENT)
        out = attn(lq, lk, lv, **kw)
    else:   # qkvpacked
        from yunchang import LongContextAttentionQKVPacked
        qkv = torch.stack([lq.detach(), lk.detach(), lv.detach()], dim=2).requires_grad_(bwd)
        attn = LongContextAttentionQKVPacked(ring_impl_type=impl, attn_type=AttnType.TORCH_EFFICIENT)
        out = attn(qkv, **kw)
    res = {"out": _bits(out)}
    if bwd:
        out.backward(ldo)
        if layer == "qkvpacked":
            res.update(dq=_bits(qkv.grad[:, :, 0]), dk=_bits(qkv.grad[:, :, 1]), dv=_bits(qkv.grad[:, :, 2]))
        else:
            res.update(dq=_bits(lq.grad), dk=_bits(lk.grad), dv=_bits(lv.grad))
    ret[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def main():
    only = set(sys.argv[1:])
    for i, case in enumerate(CASES):
        name, ws = case[0], case[1]
        if only and name.split(":")[0] not in only:
            continue
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(_worker, args=(ws, case, 29650 + i, ret), nprocs=ws, join=True)
        _, _, ud, rd, impl, B, S, Hq, Hkv, D, dtype_s, bwd = case[:12]
        causal = case[12] if len(case) > 12 else True
        layer = name.split(":")[1] if ":" in name else "hybrid"
        name = name.split(":")[0]
        blob = dict(ws=ws, ud=ud, rd=rd, impl=impl, B=B, S=S, Hq=Hq, Hkv=Hkv, D=D,
                    dtype=dtype_s, bwd=bwd, causal=causal, seed=SEED, layer=layer)
        for r in range(ws):
            for key in ("out", "dq", "dk", "dv"):
                if key in ret[r]:
                    blob[f"{key}_r{r}"] = ret[r][key]
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **blob)
        print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
