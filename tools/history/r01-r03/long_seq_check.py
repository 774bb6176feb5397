"""Long-sequence sanity (addressing, accumulation): S = 131072 tokens on one GPU, forward + backward, against
torch SDPA on the same GPU (dev tool)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F
import yunchang_amd  # noqa
from yunchang_amd.kernels import hip_attn_func
dev = torch.device("cuda:0")
for (B, S, H, Hkv, D) in ((1, 131072, 4, 4, 128), (1, 65536, 8, 2, 128)):
    g = torch.Generator(device=dev).manual_seed(3)
    q = torch.randn((B, S, H, D), device=dev, generator=g).to(torch.bfloat16)
    k = torch.randn((B, S, Hkv, D), device=dev, generator=g).to(torch.bfloat16)
    v = torch.randn((B, S, Hkv, D), device=dev, generator=g).to(torch.bfloat16)
    do = torch.randn((B, S, H, D), device=dev, generator=g).to(torch.bfloat16)
    ours = [t.clone().requires_grad_(True) for t in (q, k, v)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = hip_attn_func(*ours, causal=True)
    out.backward(do)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    rep = H // Hkv
    ref = [q.transpose(1, 2).clone().requires_grad_(True), k.repeat_interleave(rep, 2).transpose(1, 2).clone().requires_grad_(True),
           v.repeat_interleave(rep, 2).transpose(1, 2).clone().requires_grad_(True)]
    ro = F.scaled_dot_product_attention(*ref, is_causal=True)
    ro.backward(do.transpose(1, 2))
    eo = float((out.float() - ro.transpose(1, 2).float()).abs().max())
    gk = ref[1].grad.transpose(1, 2).reshape(B, S, Hkv, rep, D).float().sum(3)
    gv = ref[2].grad.transpose(1, 2).reshape(B, S, Hkv, rep, D).float().sum(3)
    errs = [float((ours[0].grad.float() - ref[0].grad.transpose(1, 2).float()).abs().max()),
            float((ours[1].grad.float() - gk).abs().max()), float((ours[2].grad.float() - gv).abs().max())]
    fl = 3.5 * 4.0 * B * H * S * S * D * 0.5
    print(f"LONG B{B} S{S} H{H}/{Hkv}: fwd+bwd {ms:.1f} ms ({fl / ms / 1e9:.0f} TFLOP/s incl. first-call overheads), "
          f"max abs err out {eo:.3e} dq {errs[0]:.3e} dk {errs[1]:.3e} dv {errs[2]:.3e}")
