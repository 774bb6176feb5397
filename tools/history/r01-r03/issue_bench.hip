// issue_bench -- how many non-MFMA instructions per v_mfma_f32_32x32x16_bf16 can a gfx950 SIMD hide?
// DEV TOOL (not part of libusp_hip.so).  Every wave runs   loop { 1 MFMA ; K VALU (T of them v_exp_f32) ; L ds_read_b128 }
// on independent registers (4 rotating accumulators, no data dependence between the fillers and the MFMAs), with W waves
// per SIMD.  Prints the MFMA throughput relative to the 2.5 PFLOP/s peak.  The flash kernels sit at W = 2 with
// ~3 VALU + ~1.3 LDS reads + ~1 wait per MFMA and per wave.
//   build: hipcc --offload-arch=gfx950 -O3 tools/issue_bench.hip -o gpurun_tools/issue_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

// X: extra per-MFMA scalar-side fillers: bit 0 = one s_waitcnt lgkmcnt(15) (never blocks), bit 1 = one s_nop 0,
// bit 2 = one SALU add
template <int K, int T, int L, int X = 0>
__global__ __launch_bounds__(512, 2) void issue_kernel(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  u32x4 a = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  if (X & 8) {            // operands with random mantissas / signs / exponents near 1 (bf16 pairs): MFMA power depends on the data
    unsigned h = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
    for (int i = 0; i < 4; ++i) {
      h = h * 1664525u + 1013904223u; a[i] = (h & 0x807f807fu) | 0x3f003f00u | ((h >> 3) & 0x00800080u);
      h = h * 1664525u + 1013904223u; b[i] = (h & 0x807f807fu) | 0x3f003f00u | ((h >> 5) & 0x00800080u);
    }
  }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = 0.001f * (lane + i);
  u32x4 ld[2] = {a, a};
  unsigned sx = iters;
  const __attribute__((address_space(3))) char* lp = (const __attribute__((address_space(3))) char*)smem + (threadIdx.x & 255) * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[m], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float& v = x[(m * K + k) & 7];
        if (k < T) asm volatile("v_exp_f32 %0, %0" : "+v"(v));
        else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v) : "v"(x[(m * K + k + 1) & 7]));
      }
#pragma unroll
      for (int l = 0; l < L; ++l) {
        u32x4& d = ld[l & 1];
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(lp), "i"(4096 * ((m * 2 + l) & 7)));
      }
      if (X & 1) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
      if (X & 2) asm volatile("s_nop 0");
      if (X & 4) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sx));
      __builtin_amdgcn_sched_barrier(0);
    }
    if (L) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += x[i];
  s += __builtin_bit_cast(float, ld[0][0]) + __builtin_bit_cast(float, ld[1][1]) + (float)sx;
  if (s == 123.456f) out[threadIdx.x] = s;
}

template <int K, int T, int L, int X = 0> static void run(float* out, int waves_per_simd) {
  const int iters = 2000, threads = 256 * waves_per_simd;
  hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  hipLaunchKernelGGL((issue_kernel<K, T, L, X>), dim3(256), dim3(threads), 40960, 0, out, iters);
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipEventRecord(e0));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((issue_kernel<K, T, L, X>), dim3(256), dim3(threads), 40960, 0, out, iters);
  HIP_OK(hipEventRecord(e1)); HIP_OK(hipEventSynchronize(e1));
  float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
  const double flops = 256.0 * (threads / 64) * iters * 4 * 32768.0;
  const double tf = flops / (ms * 1e-3) / 1e12;
  printf("W=%d waves/SIMD  per MFMA: %d VALU (%d exp) + %d ds_read_b128%s%s%s : %7.1f TFLOP/s = %4.1f %% of 2500\n",
         waves_per_simd, K, T, L, (X & 1) ? " + s_waitcnt" : "", (X & 2) ? " + s_nop" : "", (X & 4) ? " + SALU" : "", tf, tf / 25.0);
}


// Two waves per SIMD that alternate a filler-heavy and a filler-light phase of 16 MFMAs each (the forward kernel's phase A =
// 5.25 VALU incl. 1.5 exp + 1 LDS read per MFMA, phase B = 2.75 VALU incl. 0.5 exp + 2 LDS reads), one s_barrier per 32 MFMAs.
// SKEW = 0: all 8 waves run heavy, light, barrier (both waves of a SIMD are in the same phase at the same time);
// SKEW = 1: waves 4-7 (the second wave of every SIMD) run half an iteration behind: light, heavy, barrier.
template <int K, int T2, int L> __device__ __forceinline__ void phase16(f32x16 (&acc)[4], u32x4 a, u32x4 b, float (&x)[8], u32x4 (&ld)[2],
                                                                        const __attribute__((address_space(3))) char* lp) {
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[m & 3], 0, 0, 0);
    const int nexp = (T2 >> 1) + ((T2 & 1) & (m & 1));        // T2 = exps per TWO MFMAs
#pragma unroll
    for (int k = 0; k < K + ((m & 3) == 0 && K == 5 ? 1 : 0); ++k) {
      float& v = x[(m * K + k) & 7];
      if (k < nexp) asm volatile("v_exp_f32 %0, %0" : "+v"(v));
      else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v) : "v"(x[(m * K + k + 1) & 7]));
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
      u32x4& d = ld[l & 1];
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(lp), "i"(4096 * ((m * 2 + l) & 7)));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int SKEW> __global__ __launch_bounds__(512, 2) void phase_kernel(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4 a, b;
  unsigned h = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
  for (int i = 0; i < 4; ++i) {
    h = h * 1664525u + 1013904223u; a[i] = (h & 0x807f807fu) | 0x3f003f00u | ((h >> 3) & 0x00800080u);
    h = h * 1664525u + 1013904223u; b[i] = (h & 0x807f807fu) | 0x3f003f00u | ((h >> 5) & 0x00800080u);
  }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = 0.001f * (lane + i);
  u32x4 ld[2] = {a, a};
  const __attribute__((address_space(3))) char* lp = (const __attribute__((address_space(3))) char*)smem + (threadIdx.x & 255) * 16;
  if (SKEW && wave >= 4) {
    phase16<5, 3, 1>(acc, a, b, x, ld, lp);
    __builtin_amdgcn_s_barrier();
    for (int it = 1; it < iters; ++it) {
      phase16<3, 1, 2>(acc, a, b, x, ld, lp);
      phase16<5, 3, 1>(acc, a, b, x, ld, lp);
      __builtin_amdgcn_s_barrier();
    }
    phase16<3, 1, 2>(acc, a, b, x, ld, lp);
  } else {
    for (int it = 0; it < iters; ++it) {
      phase16<5, 3, 1>(acc, a, b, x, ld, lp);
      phase16<3, 1, 2>(acc, a, b, x, ld, lp);
      __builtin_amdgcn_s_barrier();
    }
  }
  float sum = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
  for (int i = 0; i < 8; ++i) sum += x[i];
  sum += __builtin_bit_cast(float, ld[0][0]) + __builtin_bit_cast(float, ld[1][1]);
  if (sum == 123.456f) out[threadIdx.x] = sum;
}

template <int SKEW> static void run_phase(float* out) {
  const int iters = 500, threads = 512, reps = 20;
  hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((phase_kernel<SKEW>), dim3(256), dim3(threads), 40960, 0, out, iters);
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((phase_kernel<SKEW>), dim3(256), dim3(threads), 40960, 0, out, iters);
  HIP_OK(hipEventRecord(e1)); HIP_OK(hipEventSynchronize(e1));
  float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double flops = 256.0 * 8 * iters * 32 * 32768.0;
  printf("phases (heavy 5.25 VALU/1.5 exp/1 LDS | light 2.75/0.5/2), random operands, barrier per 32 MFMAs, %s: %7.1f TFLOP/s = %4.1f %%\n",
         SKEW ? "waves 4-7 HALF AN ITERATION BEHIND" : "all waves in phase", flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 25.0);
}

// The dQ kernel's tile: 48 MFMAs with 144 VALU (32 exp) of element work.  SHAPE 0: as scheduled today -- 16 bare, 16 with 4.5 VALU
// (1 exp), 8 with 9 VALU (2 exp), 8 bare; SHAPE 1: the same work spread evenly, 3 VALU (2 exps per 3 MFMAs) behind every MFMA.
// LDS: 1 ds_read_b128 per MFMA in the first 32, 2 in the last 16 (transpose reads come in pairs).
template <int NM, int K, int T3, int L> __device__ __forceinline__ void phaseN(f32x16 (&acc)[4], u32x4 a, u32x4 b, float (&x)[8], u32x4 (&ld)[2],
                                                                               const __attribute__((address_space(3))) char* lp) {
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[m & 3], 0, 0, 0);
    const int nexp = T3 / 3 + ((m % 3) < (T3 % 3) ? 1 : 0);    // T3 = exps per THREE MFMAs
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float& v = x[(m * K + k) & 7];
      if (k < nexp) asm volatile("v_exp_f32 %0, %0" : "+v"(v));
      else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v) : "v"(x[(m * K + k + 1) & 7]));
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
      u32x4& d = ld[l & 1];
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(lp), "i"(4096 * ((m * 2 + l) & 7)));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int SHAPE> __global__ __launch_bounds__(512, 2) void dq_shape_kernel(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  u32x4 a, b;
  unsigned h = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
  for (int i = 0; i < 4; ++i) {
    h = h * 1664525u + 1013904223u; a[i] = (h & 0x807f807fu) | 0x3f003f00u | ((h >> 3) & 0x00800080u);
    h = h * 1664525u + 1013904223u; b[i] = (h & 0x807f807fu) | 0x3f003f00u | ((h >> 5) & 0x00800080u);
  }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = 0.001f * (lane + i);
  u32x4 ld[2] = {a, a};
  const __attribute__((address_space(3))) char* lp = (const __attribute__((address_space(3))) char*)smem + (threadIdx.x & 255) * 16;
  for (int it = 0; it < iters; ++it) {
    if (SHAPE == 0) {
      phaseN<16, 0, 0, 1>(acc, a, b, x, ld, lp);
      phaseN<16, 5, 3, 1>(acc, a, b, x, ld, lp);      // 4.5 -> alternate 4 / 5: use 5,4 below
      phaseN<8, 8, 6, 2>(acc, a, b, x, ld, lp);
      phaseN<8, 0, 0, 2>(acc, a, b, x, ld, lp);
    } else {
      phaseN<32, 3, 2, 1>(acc, a, b, x, ld, lp);
      phaseN<16, 3, 2, 2>(acc, a, b, x, ld, lp);
    }
    __builtin_amdgcn_s_barrier();
  }
  float sum = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
  for (int i = 0; i < 8; ++i) sum += x[i];
  sum += __builtin_bit_cast(float, ld[0][0]) + __builtin_bit_cast(float, ld[1][1]);
  if (sum == 123.456f) out[threadIdx.x] = sum;
}

template <int SHAPE> static void run_dq_shape(float* out) {
  const int iters = 400, threads = 512, reps = 20;
  hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((dq_shape_kernel<SHAPE>), dim3(256), dim3(threads), 40960, 0, out, iters);
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((dq_shape_kernel<SHAPE>), dim3(256), dim3(threads), 40960, 0, out, iters);
  HIP_OK(hipEventRecord(e1)); HIP_OK(hipEventSynchronize(e1));
  float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double flops = 256.0 * 8 * iters * 48 * 32768.0;
  printf("dQ tile (48 MFMAs, ~144 VALU incl. 32 exp), random operands, %s: %7.1f TFLOP/s = %4.1f %%\n",
         SHAPE ? "fillers spread EVENLY (3 per MFMA)" : "as scheduled today (0 | 5 | 8 | 0 per MFMA)", flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 25.0);
}

// sustained clocks: the same loop for ~2.5 s, one figure per 250 ms window
template <int K, int T, int L, int X = 0> static void sustain(float* out) {
  const int iters = 2000, threads = 512, per_window = 200;
  const double flops = 256.0 * (threads / 64) * iters * 4 * 32768.0 * per_window;
  hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  printf("sustained, W=2, %s operands, per MFMA %d VALU (%d exp) + %d ds_read_b128, TFLOP/s per window of %d launches:",
         (X & 8) ? "RANDOM" : "constant", K, T, L, per_window);
  double t_total = 0;
  while (t_total < 2500.0) {
    HIP_OK(hipEventRecord(e0));
    for (int i = 0; i < per_window; ++i) hipLaunchKernelGGL((issue_kernel<K, T, L, X>), dim3(256), dim3(threads), 40960, 0, out, iters);
    HIP_OK(hipEventRecord(e1)); HIP_OK(hipEventSynchronize(e1));
    float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    t_total += ms;
    printf(" %.0f", flops / (ms * 1e-3) / 1e12);
  }
  printf("\n");
}

int main(int argc, char** argv) {
  float* out; HIP_OK(hipMalloc(&out, 4096));
  if (argc > 1 && argv[1][0] == 'd') {      // issue_bench dq
    for (int r = 0; r < 3; ++r) { run_dq_shape<0>(out); run_dq_shape<1>(out); }
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'p') {      // issue_bench phase
    for (int r = 0; r < 3; ++r) { run_phase<0>(out); run_phase<1>(out); }
    return 0;
  }
  if (argc > 1) {      // issue_bench sustain
    sustain<0, 0, 0>(out); sustain<0, 0, 0, 8>(out); sustain<3, 1, 1>(out); sustain<3, 1, 1, 8>(out); sustain<4, 1, 2, 8>(out);
    return 0;
  }
  for (int w = 1; w <= 2; ++w) {
    run<0, 0, 0>(out, w); run<1, 0, 0>(out, w); run<2, 0, 0>(out, w); run<3, 0, 0>(out, w); run<4, 0, 0>(out, w);
    run<6, 0, 0>(out, w); run<8, 0, 0>(out, w);
    run<3, 1, 0>(out, w); run<4, 1, 1>(out, w); run<3, 1, 1>(out, w); run<3, 1, 2>(out, w); run<6, 2, 2>(out, w);
    run<0, 0, 1>(out, w); run<0, 0, 2>(out, w);
    run<3, 1, 1, 1>(out, w); run<3, 1, 1, 2>(out, w); run<3, 1, 1, 4>(out, w); run<3, 1, 1, 7>(out, w);
    run<3, 0, 1>(out, w); run<2, 1, 1>(out, w); run<2, 0, 1>(out, w);
  }
  return 0;
}
