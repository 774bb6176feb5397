// atomic_bench -- how fast can gfx950 absorb the fp32 atomic traffic of a fused flash backward (dQ partials)?
// DEV TOOL (not part of libusp_hip.so).  Emulates the access pattern: 256 persistent workgroups of 8 waves walk
// (batch*head, 128-key block) items; per 64-row query tile every wave adds a [64 q][16 d] fp32 patch (16 registers,
// MFMA 16x16 result layout: lane l, reg 4*qb + r -> q = 16*qb + 4*(l>>4) + r, d = 16*wave + (l&15)) into
// dq[b, q, h, :].  Causal item lengths, C2 shape (B2 S8192 H16 D128): 133,120 tiles x 32 KiB = 4.36 GB of adds.
//   build: hipcc --offload-arch=gfx950 -O3 tools/atomic_bench.hip -o gpurun_tools/atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

// MODE 0: agent-scope atomic add   1: workgroup-scope atomic add (executes in this XCD's L2)
// MODE 2: plain load+add+store (NOT a valid reduction; bandwidth reference)   3: plain store only
// LAYOUT 0: 16x16 result layout (4 q rows x 64 B per instruction)   1: 32x32 layout (2 q rows x 128 B per instruction)
template <int MODE, int LAYOUT>
__global__ __launch_bounds__(512) void dq_traffic(float* dq, int B, int S, int H, int nkb, int by_xcd) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_items = B * H * nkb;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3, wgs_l = gridDim.x >> 3, items_l = n_items >> 3;
  for (int pass = 0;; ++pass) {
    int item;
    if (by_xcd) {
      const int li = pass * wgs_l + ((pass & 1) ? wgs_l - 1 - loc : loc);
      if (li >= items_l) break;
      item = xcd * items_l + li;
    } else {
      item = pass * gridDim.x + blockIdx.x;
      if (item >= n_items) break;
    }
    const int kb = item % nkb, bh = item / nkb;
    const int h = bh % H, b = bh / H;
    const int t0 = (kb * 128) / 64, t1 = S / 64;
    for (int t = t0; t < t1; ++t) {
      float* base = dq + ((size_t)(b * S + t * 64) * H + h) * 128;
      const size_t rs = (size_t)H * 128;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float* p;
        if (LAYOUT == 0) {
          const int q = 16 * (r >> 2) + 4 * (lane >> 4) + (r & 3);
          p = base + q * rs + 16 * wave + (lane & 15);
        } else {   // wave = (q half, d block of 32): q = 32*(wave&1) + (r&3) + 8*(r>>2) + 4*(lane>>5), d = 32*(wave>>1) + (lane&31)
          const int q = 32 * (wave & 1) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          p = base + q * rs + 32 * (wave >> 1) + (lane & 31);
        }
        const float v = 1e-3f * (r + 1);
        if (MODE == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 1) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 2) *p = *p + v;
        else *p = v;
      }
    }
  }
}

template <int MODE, int LAYOUT> static float run(float* dq, int B, int S, int H, int by_xcd, int iters) {
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  const int nkb = S / 128;
  hipLaunchKernelGGL((dq_traffic<MODE, LAYOUT>), dim3(256), dim3(512), 0, 0, dq, B, S, H, nkb, by_xcd);
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((dq_traffic<MODE, LAYOUT>), dim3(256), dim3(512), 0, 0, dq, B, S, H, nkb, by_xcd);
  HIP_OK(hipEventRecord(e1));
  HIP_OK(hipEventSynchronize(e1));
  float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

int main(int argc, char** argv) {
  const int B = 2, S = argc > 1 ? atoi(argv[1]) : 8192, H = 16;
  const size_t n = (size_t)B * S * H * 128;
  float* dq; HIP_OK(hipMalloc(&dq, n * 4)); HIP_OK(hipMemset(dq, 0, n * 4));
  const int nkb = S / 128;
  double tiles = 0; for (int kb = 0; kb < nkb; ++kb) tiles += S / 64 - kb * 2;
  const double gb = tiles * B * H * 32768.0 / 1e9;
  printf("dQ partial traffic at B%d S%d H%d D128 causal: %.2f GB of fp32 adds (dq tensor %.1f MB)\n", B, S, H, gb, n * 4 / 1e6);
  const char* names[4] = {"atomic agent", "atomic workgroup(L2)", "plain rmw", "plain store"};
#define RUN(M, L) for (int x = 1; x >= 0; --x) { float ms = run<M, L>(dq, B, S, H, x, 5); \
    printf("%-22s layout %s  %s : %.3f ms  %.2f TB/s\n", names[M], L ? "32x32" : "16x16", x ? "xcd-contiguous" : "round-robin   ", ms, gb / ms); }
  RUN(0, 0) RUN(0, 1) RUN(1, 0) RUN(1, 1) RUN(2, 0) RUN(3, 0) RUN(3, 1)
  // correctness of the L2-local atomics when every contributor to a row sits on one XCD (xcd-contiguous walk):
  HIP_OK(hipMemset(dq, 0, n * 4));
  hipLaunchKernelGGL((dq_traffic<1, 0>), dim3(256), dim3(512), 0, 0, dq, B, S, H, nkb, 1);
  HIP_OK(hipDeviceSynchronize());
  float* h = (float*)malloc(n * 4); HIP_OK(hipMemcpy(h, dq, n * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (int b = 0; b < B; ++b) for (int q = 0; q < S; ++q) for (int hh = 0; hh < H; ++hh) for (int d = 0; d < 128; ++d) {
    // row q of a tile receives one add per key block kb with kb*128/64 <= q/64, value 1e-3*(r+1), r = 4*((q%64)/16) + (q%4)
    const int r = 4 * ((q % 64) / 16) + (q % 4);
    const int cnt = (q / 64) / 2 + 1;
    const float want = cnt * 1e-3f * (r + 1);
    const float got = h[((size_t)(b * S + q) * H + hh) * 128 + d];
    if (fabsf(got - want) > 1e-3f * want + 1e-6f) ++bad;
  }
  printf("workgroup-scope atomics, xcd-contiguous walk: %zu wrong of %zu\n", bad, n);
  return 0;
}
