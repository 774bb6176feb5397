"""The reference's TORCH attention path on a HOST, restated for bench.py's ``cpu_baseline`` leg.

TEST / BASELINE INFRASTRUCTURE ONLY (like everything under oracle/): nothing in the product package imports it.

/root/reference does not exist on the GPU box, so the reference cannot be imported there; this file restates, call
for call, what `LongContextAttention(attn_type=AttnType.TORCH_EFFICIENT)` executes on ONE rank (world size 1 =
BASELINE.json configs[1]) for a forward pass:

  * yunchang/hybrid/attn_layer.py:111-119,156-158 -- SeqAllToAll4D on q, k, v and on the output.  At ulysses
    degree 1 `all_to_all_4D` moves nothing between ranks but still makes its two layout copies
    (yunchang/comm/all_to_all.py:39-49,62-65): reshape -> transpose -> .contiguous(), then back;
  * yunchang/ring/ring_flash_attn.py:20-57 -- one ring step = one block call, then `update_out_and_lse`
    adopting the block (ring/utils.py:38-42: out -> fp32, lse -> (B,S,H,1)) and the final `.to(q.dtype)`;
  * yunchang/kernels/attention.py:44-136 `pytorch_attn_forward(op_type="efficient")` -- (B,S,H,D) -> (B,H,S,D)
    views, the aten op, transpose back, LSE cast to q.dtype (:135).  The aten op itself,
    `_scaled_dot_product_efficient_attention`, has NO CPU kernel; on a host the path
    can only run with `aten::_scaled_dot_product_flash_attention_for_cpu` in its place, which is the
    substitution SURVEY.md Appendix A verified against the reference and tests/golden/make_golden.py uses.
"""
import torch


def pytorch_attn_forward_cpu(q, k, v, softmax_scale=None, causal=False):
    """kernels/attention.py:44-136 with the CPU flash op in the efficient op's place."""
    qt, kt, vt = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)           # :62-64
    out, lse = torch.ops.aten._scaled_dot_product_flash_attention_for_cpu(qt, kt, vt, 0.0, causal,
                                                                           scale=softmax_scale)[:2]
    return out.transpose(1, 2), lse.to(q.dtype)                                     # :131-135


def _all_to_all_4d_single_rank(x, scatter_idx, gather_idx):
    """comm/all_to_all.py:36-67 / :69-102 at world size 1: the collective is the identity, the copies stay."""
    bs, s, h, d = x.shape
    if scatter_idx == 2 and gather_idx == 1:
        t = x.reshape(bs, s, 1, h, d).transpose(0, 2).contiguous()                  # :45-49
        return t.reshape(s, bs, h, d).transpose(0, 1).contiguous().reshape(bs, s, h, d)    # :62-65
    t = x.reshape(bs, 1, s, h, d).transpose(0, 3).transpose(0, 1).contiguous().reshape(1, h, s, bs, d)   # :80-84
    return t.reshape(h, s, bs, d).transpose(0, 2).contiguous().reshape(bs, s, h, d)        # :98-100


def long_context_attention_forward_cpu(q, k, v, causal=True, softmax_scale=None):
    """One forward of the reference's hybrid layer on one rank, host tensors (B,S,H,D)."""
    if softmax_scale is None:
        softmax_scale = q.shape[-1] ** -0.5
    q, k, v = (_all_to_all_4d_single_rank(t, 2, 1) for t in (q, k, v))              # attn_layer.py:111-119
    block_out, block_lse = pytorch_attn_forward_cpu(q, k, v, softmax_scale, causal) # ring_flash_attn.py:36-48
    out = block_out.to(torch.float32)                                               # ring/utils.py:38-42
    lse = block_lse.transpose(-2, -1).unsqueeze(dim=-1)
    out = out.to(q.dtype)                                                           # ring_flash_attn.py:55
    lse = lse.squeeze(dim=-1).transpose(1, 2)                                       # :56
    return _all_to_all_4d_single_rank(out, 1, 2), lse                               # attn_layer.py:156-158
