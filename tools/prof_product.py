"""The PRODUCT path under the profiler (DEV / measurement helper; run under rocprofv3 by tools/prof_round.sh).

    python tools/prof_product.py c2|c5|layer [iters]

  c2     forward + backward kernels at BASELINE configs[1] (B2 S8192 H16 D128 bf16 causal) through yunchang_amd._C exactly
         as bench.py's roofline.fwd_bwd times them: 16-bit final outputs (dq16 / dk16 / dv16), delta launch included
  c5     the same at the metric's shape on one GPU (B1 S65536 H32/Hkv4)
  layer  LongContextAttention.forward + out.backward through autograd at C2 (1 x 1 grid)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "c2"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if what == "layer":
        import torch.distributed as dist
        import yunchang_amd as Y
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29761")
        dist.init_process_group("nccl", rank=0, world_size=1)
        Y.set_seq_parallel_pg(1, 1, 0, 1)
        ms = bench._layer_fwd_bwd(2, 8192, 16, 16, 128, dev, iters or 20)
        print(f"layer fwd+bwd C2: {ms:.4f} ms per step")
        w = bench.WORKLOADS[1]
        ms = bench._layer_fwd_bwd(w["B"], w["S"], w["Hq"], w["Hkv"], w["D"], dev, 3, warm=1)
        print(f"layer fwd+bwd at the N=1 workload (B1 S65536 H32/Hkv4): {ms:.3f} ms per step")
        dist.destroy_process_group()
        return
    c = bench.C2 if what == "c2" else bench.WORKLOADS[1]
    n = iters or (40 if what == "c2" else 3)
    if what == "c2":
        bench._fwd_bwd_kernels(c["B"], c["S"], c["Hq"], c["Hkv"], c["D"], dev, 200)      # sustained clocks first
    t = bench._fwd_bwd_kernels(c["B"], c["S"], c["Hq"], c["Hkv"], c["D"], dev, n)
    print(f"{what}: fwd {t['fwd_ms']} ms ({t['fwd']:.1f} TFLOP/s)  bwd {t['bwd_ms']} ms ({t['bwd']:.1f} TFLOP/s)")


if __name__ == "__main__":
    main()
