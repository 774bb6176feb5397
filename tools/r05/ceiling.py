"""What limits the flash kernels on THIS part: the MFMA-only ceiling on the bench's data, with clock and power telemetry
(DEV / measurement tool, round 5; run on the GPU box: `python tools/r05/ceiling.py > gpurun_out/r05/ceiling.txt`).

Phases of ~2.5 s each, back to back on one box in one process, a sampler thread reading board power and shader clock
(sysfs hwmon: power1_average / power1_input, freq1_input; rocm-smi as fallback) every 50 ms:
    idle | MFMA-only loop on N(0,1) bf16, 1 and 2 waves per SIMD | the same on zeros | forward kernel (C2, 64K) |
    dK/dV kernel | dQ kernel
Per phase: executed TFLOP/s (MFMA FLOPs actually issued: forward 2 matmuls, dK/dV 4, dQ 3), the loop's own shader clock
where the kernel records it (usp_mfma_probe: s_memtime / s_memrealtime), mean sampled clock and power.  The product kernels'
MFMA-pipe occupancy comes from the rocprofv3 PMC passes of the same box (tools/prof_round.sh -> profiles/r05_rocprof_summary.txt)."""
import glob
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.stop = [], False
        hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        self.power = next((p for h in hw for p in (h + "/power1_average", h + "/power1_input") if os.path.exists(p)), None)
        self.freq = next((h + "/freq1_input" for h in hw if os.path.exists(h + "/freq1_input")), None)
        self.src = "sysfs hwmon" if (self.power or self.freq) else "rocm-smi"

    def read(self):
        if self.power or self.freq:
            def rd(p):
                try:
                    return float(open(p).read().strip())
                except Exception:
                    return float("nan")
            return (rd(self.power) / 1e6 if self.power else float("nan"), rd(self.freq) / 1e9 if self.freq else float("nan"))
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
            lines = [l for l in out.splitlines() if l and not l.startswith("WARNING")]
            hdr, val = lines[0].split(","), lines[1].split(",")
            d = dict(zip(hdr, val))
            pw = next((float(v) for k, v in d.items() if "ower" in k and v.replace(".", "", 1).isdigit()), float("nan"))
            ck = next((float(v.strip("()Mhz")) / 1e3 for k, v in d.items() if "sclk" in k.lower() and "(" in v), float("nan"))
            return pw, ck
        except Exception:
            return float("nan"), float("nan")

    def run(self):
        while not self.stop:
            pw, ck = self.read()
            self.rows.append((time.perf_counter(), pw, ck))
            time.sleep(0.05 if self.src != "rocm-smi" else 0.4)

    def mean(self, t0, t1):
        sel = [(p, c) for t, p, c in self.rows if t0 + 0.3 <= t <= t1]          # skip the first 0.3 s of a phase (ramp)
        f = lambda xs: sum(xs) / len(xs) if xs else float("nan")
        return f([p for p, _ in sel if p == p]), f([c for _, c in sel if c == c]), len(sel)


def phase(name, fn, seconds, sampler, flops_per_call, clocks=None):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            fn()
        n += 8
        torch.cuda.current_stream().synchronize() if n % 64 == 0 else None
    e1.record()
    e1.synchronize()
    t1 = time.perf_counter()
    ms = e0.elapsed_time(e1) / max(n, 1)
    pw, ck, ns = sampler.mean(t0, t1)
    own = ""
    if clocks is not None:
        c = clocks.tolist()
        own = f"{c[0] / max(1, c[1]) * 0.1:5.3f}"
    tf = flops_per_call / (ms * 1e-3) / 1e12 if flops_per_call else 0.0
    print(f"{name:58s} {ms:9.4f} ms  {tf:7.1f} TFLOP/s executed  loop clock {own or '  -  ':>5s} GHz  sampled {ck:5.2f} GHz  {pw:6.0f} W  ({ns} samples)")
    return tf


def main():
    from yunchang_amd import _C
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    s = Sampler()
    s.start()
    print(f"telemetry: {s.src} (power: {s.power}, clock: {s.freq}); device: {torch.cuda.get_device_name(0)}")
    print("phase" + " " * 53 + "per call    executed rate            clock of the loop   sampled clock / board power")
    time.sleep(1.5)
    t0 = time.perf_counter()
    time.sleep(1.5)
    pw, ck, ns = s.mean(t0 - 0.3, time.perf_counter())
    print(f"{'idle':58s} {'':9s}     {'':7s}                         {'':5s}      sampled {ck:5.2f} GHz  {pw:6.0f} W  ({ns} samples)")
    g = torch.Generator(device=dev).manual_seed(3)
    normal = torch.randn(1 << 20, device=dev, generator=g).to(torch.bfloat16)
    zeros = torch.zeros(1 << 20, device=dev, dtype=torch.bfloat16)
    clocks = torch.zeros(2, dtype=torch.int64, device=dev)
    res = {}
    for name, buf in (("N(0,1) bf16", normal), ("zeros", zeros)):
        for w in (1, 2):
            iters = 2000 // w
            fl = [0.0]

            def f():
                fl[0] = _C.mfma_probe(buf, iters, w, clocks)
            f()
            res[(name, w)] = phase(f"MFMA-only loop, {name}, {w} wave(s) per SIMD", f, 2.5, s, fl[0], clocks)
    top = max(res[("N(0,1) bf16", 1)], res[("N(0,1) bf16", 2)])
    # the product kernels, C2 and the N = 1 workload's shape
    for label, (B, S, Hq, Hkv) in (("C2 B2 S8192 H16", (2, 8192, 16, 16)), ("B1 S65536 H32/Hkv4", (1, 65536, 32, 4))):
        D = 128
        q, k, v, do = bench._kernel_inputs(B, S, Hq, Hkv, D, dev)
        out = torch.empty_like(q)
        lse = torch.empty((B, Hq, S), device=dev, dtype=torch.float32)
        delta = torch.empty_like(lse)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        sc = D ** -0.5
        mm = 0.5 * bench.fwd_flops(B, Hq, S, D)
        _C.flash_fwd(q, k, v, sc, True, lse, out)
        _C.bwd_delta(do, out, delta)
        secs = 2.5 if S < 65536 else 1.5
        a = (do, q, k, v, lse, delta, None, None, None, sc, True)
        kw = dict(dq16=dq, dk16=dk, dv16=dv)
        for name, fn, n_mm in ((f"forward kernel (flash_fwd64), {label}", lambda: _C.flash_fwd(q, k, v, sc, True, lse, out), 2),
                               (f"dK/dV kernel (flash_bwd_dkdv64), {label}", lambda: _C.flash_bwd(*a, only="dkdv", **kw), 4),
                               (f"dQ kernel (flash_bwd_dq64), {label}", lambda: _C.flash_bwd(*a, only="dq", **kw), 3)):
            tf = phase(name, fn, secs, s, n_mm * mm)
            print(f"{'':58s} = {tf / top:5.3f} of the MFMA-only ceiling on N(0,1) ({top:.0f}), {tf / 2500:5.3f} of the nominal 2500")
        del q, k, v, do, out, lse, delta, dq, dk, dv
    s.stop = True


if __name__ == "__main__":
    main()
