"""Does the third-party op behind the reference's TORCH_EFFICIENT path (kernels/attention.py:76-86) run on this
box, and how fast?  (dev probe)"""
import torch, time
dev = torch.device("cuda:0")
B, S, H, D = 2, 8192, 16, 128
torch.manual_seed(0)
q, k, v = (torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16) for _ in range(3))
qt, kt, vt = (t.transpose(1, 2) for t in (q, k, v))
fl = 4.0 * B * H * S * S * D * 0.5


def timeit(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, fn in (
    ("aten efficient", lambda: torch.ops.aten._scaled_dot_product_efficient_attention(qt, kt, vt, None, True, 0.0, True, scale=D ** -0.5)),
    ("aten flash", lambda: torch.ops.aten._scaled_dot_product_flash_attention(qt, kt, vt, 0.0, True, False, scale=D ** -0.5)),
):
    try:
        r = fn()
        ms = timeit(fn)
        print(f"ATEN {name}: ok, out {tuple(r[0].shape)} {r[0].dtype}, lse {tuple(r[1].shape)} {r[1].dtype}, {ms:.3f} ms, {fl / ms / 1e9:.1f} TFLOP/s")
    except Exception as e:
        print(f"ATEN {name}: FAILED {type(e).__name__}: {str(e)[:200]}")
