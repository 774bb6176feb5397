#!/usr/bin/env python3
"""Golden fixtures for the packed variable-length ring schedules, produced by
running the REFERENCE's own zigzag_ring_flash_attn_varlen_func / ring_flash_attn_varlen_func
(yunchang/ring/zigzag_ring_flash_attn_varlen.py, ring_flash_attn_varlen.py; imported read-only from
/root/reference) on CPU under torch.distributed/gloo.  Build container only.

    python tests/golden/make_golden_varlen.py      # rewrites tests/golden/v_*.npz

The reference's schedule code (get_half_index, get_half_lse, the step loop, RingComm,
update_out_and_lse, flatten/unflatten_varlen_lse) runs UNMODIFIED.  Substituted seams:
  (1) flash_attn's _flash_attn_varlen_forward / _flash_attn_varlen_backward (flash-attn is not
      installed; the reference imports them only `if HAS_FLASH_ATTN`, :4-8) -> per-sequence CPU flash
      op / fp32 torch block backward with flash-attn 2.6's argument order and return tuple
      (out at [0], padded (num_seq, H, max_seqlen) softmax_lse at [5]).
  (2) the reference prefers its Triton flatten/unflatten_varlen_lse (ring/triton_utils.py) when triton
      imports; those kernels need a GPU, so the module is pointed at the reference's OWN torch versions
      (ring/utils.py:96-117), which is its documented fallback (:14-24).
Per-rank inputs are the zigzag (resp. contiguous) shard of every sequence, exactly what the upstream
ring-flash-attention tests feed these functions; yunchang ships no extractor for packed batches, the
one used here is oracle.usp_oracle.zigzag_extract_local_varlen / basic_extract_local_varlen.
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

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from make_golden import REF, SEED, _bits, _block_bwd_torch      # noqa: E402
from oracle import usp_oracle as O                               # noqa: E402

# name, ws, impl, global sequence lengths, Hq, Hkv, D, dtype
CASES = [
    ("v_w4_zigzag_bf16", 4, "zigzag", (256, 384, 128), 2, 2, 64, "bfloat16"),
    ("v_w2_zigzag_gqa_bf16", 2, "zigzag", (64, 192, 320, 8), 4, 2, 128, "bfloat16"),
    ("v_w2_zigzag_oneseq_fp16", 2, "zigzag", (256,), 2, 2, 64, "float16"),     # slice path (:28-32)
    ("v_w2_basic_bf16", 2, "basic", (130, 62, 256), 2, 2, 64, "bfloat16"),
]


def make_inputs_varlen(lens, Hq, Hkv, D, seed=SEED):
    """Global packed q,k,v,dout (T,H,D) as float32 numpy N(0,1); tests regenerate them with this."""
    T = int(sum(lens))
    rs = np.random.RandomState(seed)
    q = rs.standard_normal((T, Hq, D)).astype(np.float32)
    k = rs.standard_normal((T, Hkv, D)).astype(np.float32)
    v = rs.standard_normal((T, Hkv, D)).astype(np.float32)
    dout = rs.standard_normal((T, Hq, D)).astype(np.float32)
    return q, k, v, dout


def _varlen_fwd(q, k, v, cu_q, cu_k, max_q, max_k, dropout_p, softmax_scale, causal=False,
                window_size=(-1, -1), softcap=0.0, alibi_slopes=None, return_softmax=False, **kw):
    n, H = len(cu_q) - 1, q.shape[1]
    g = H // k.shape[1]
    out = torch.empty_like(q)
    lse = torch.zeros(n, H, int(max_q), dtype=torch.float32)
    for i in range(n):
        qs = q[int(cu_q[i]):int(cu_q[i + 1])].transpose(0, 1)[None]
        ks = k[int(cu_k[i]):int(cu_k[i + 1])].repeat_interleave(g, 1).transpose(0, 1)[None]
        vs = v[int(cu_k[i]):int(cu_k[i + 1])].repeat_interleave(g, 1).transpose(0, 1)[None]
        assert not causal or qs.shape[2] == ks.shape[2]
        o, l_ = torch.ops.aten._scaled_dot_product_flash_attention_for_cpu(
            qs, ks, vs, dropout_p, causal, scale=softmax_scale)[:2]
        out[int(cu_q[i]):int(cu_q[i + 1])] = o[0].transpose(0, 1)
        lse[i, :, :qs.shape[2]] = l_[0]
    return out, None, None, None, None, lse, None, None


def _varlen_bwd(dout, q, k, v, out, softmax_lse, dq, dk, dv, cu_q, cu_k, max_q, max_k, dropout_p,
                softmax_scale, causal, window_size, softcap, alibi_slopes, deterministic,
                rng_state=None, **kw):
    for i in range(len(cu_q) - 1):
        a, b = int(cu_q[i]), int(cu_q[i + 1])
        c, d = int(cu_k[i]), int(cu_k[i + 1])
        _block_bwd_torch(dout[a:b][None], q[a:b][None], k[c:d][None], v[c:d][None], out[a:b][None],
                         softmax_lse[i, :, :b - a][None], dq[a:b][None], dk[c:d][None], dv[c:d][None],
                         dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes,
                         deterministic)


def _worker(rank, ws, case, port, ret):
    name, _, impl, lens, Hq, Hkv, D, dtype_s = case
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    torch.set_num_threads(1)
    sys.path.insert(0, REF)
    import yunchang.ring.utils as U
    if impl == "zigzag":
        import yunchang.ring.zigzag_ring_flash_attn_varlen as Z
        fn = Z.zigzag_ring_flash_attn_varlen_func
        extract = O.zigzag_extract_local_varlen
    else:
        import yunchang.ring.ring_flash_attn_varlen as Z
        fn = Z.ring_flash_attn_varlen_func
        extract = O.basic_extract_local_varlen
        Z.HAS_FLASH_ATTN = True          # `assert HAS_FLASH_ATTN` guards the substituted call (:55,:123)
    Z._flash_attn_varlen_forward = _varlen_fwd
    Z._flash_attn_varlen_backward = _varlen_bwd
    Z.flatten_varlen_lse = U.flatten_varlen_lse
    Z.unflatten_varlen_lse = U.unflatten_varlen_lse

    dtype = getattr(torch, dtype_s)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    q, k, v, dout = make_inputs_varlen(lens, Hq, Hkv, D)
    lq, lk, lv, ldo = (torch.from_numpy(np.ascontiguousarray(extract(t, cu, rank, ws))).to(dtype)
                       for t in (q, k, v, dout))
    lq.requires_grad_(True); lk.requires_grad_(True); lv.requires_grad_(True)
    local_cu = torch.tensor(cu // ws, dtype=torch.int32)
    max_seqlen = int(max(lens)) // ws
    out, lse, _ = fn(lq, lk, lv, local_cu, max_seqlen, dropout_p=0.0, causal=True,
                     window_size=(-1, -1), alibi_slopes=None, deterministic=False,
                     return_attn_probs=True, group=dist.group.WORLD)
    out.backward(ldo)
    # only the valid part of the padded (num_seq, H, max_seqlen) lse is defined (utils.py:110 torch.empty)
    lse_flat = U.flatten_varlen_lse(lse.detach(), local_cu)           # (H, T_local)
    ret[rank] = dict(out=_bits(out), lse=lse_flat.numpy().astype(np.float32), dq=_bits(lq.grad),
                     dk=_bits(lk.grad), dv=_bits(lv.grad))
    dist.barrier()
    dist.destroy_process_group()


def main():
    only = set(sys.argv[1:])
    for i, case in enumerate(CASES):
        name, ws, impl, lens, Hq, Hkv, D, dtype_s = case
        if only and name not in only:
            continue
        ret = mp.Manager().dict()
        mp.spawn(_worker, args=(ws, case, 29750 + i, ret), nprocs=ws, join=True)
        blob = dict(ws=ws, impl=impl, lens=np.asarray(lens, np.int64), Hq=Hq, Hkv=Hkv, D=D,
                    dtype=dtype_s, causal=True, seed=SEED, layer="ring_varlen")
        for r in range(ws):
            for key, val in ret[r].items():
                blob[f"{key}_r{r}"] = val
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **blob)
        print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
