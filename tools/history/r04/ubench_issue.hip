// DEV microbenchmark (round 4): what one wave per SIMD pays for the fillers between its MFMAs -- in particular v_exp_f32.
// One workgroup of 4 waves (one per SIMD); each mode runs ITER iterations of a 4-gap body:  per gap one
// v_mfma_f32_32x32x16_bf16 (4 independent accumulators) followed by NE v_exp_f32 and NF v_fma_f32 on independent registers.
// Prints shader cycles per gap (s_memtime).    hipcc --offload-arch=gfx950 -O2 ubench_issue.hip -o ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MF, int NE, int NF, int NS, int NL, int GRP = 1, int ACCA = 0, int NN = 0, int NW = 0, int NC = 0>
__global__ __launch_bounds__(256, 1) void k(uint64_t* out, float seed, int iters) {
  extern __shared__ char lds[];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = seed * r;
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  float e[8], f[8];
  for (int i = 0; i < 8; ++i) { e[i] = seed - i; f[i] = seed + i; }
  int s0 = 7, s1 = 1;
  u32x4 ld[4];
  const int laddr = (threadIdx.x & 63) * 16;
  for (int i = 0; i < 4; ++i) ld[i] = a;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g0 = 0; g0 < 4; g0 += GRP) {
#pragma unroll
      for (int g = g0; g < g0 + GRP; ++g) {
        if (MF && !ACCA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[g]) : "v"(a), "v"(b));
        if (MF && ACCA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[g]) : "v"(a), "v"(b));
      }
#pragma unroll
      for (int g = g0; g < g0 + GRP; ++g) {
#pragma unroll
        for (int i = 0; i < NE; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(e[(g * NE + i) & 7]));
#pragma unroll
        for (int i = 0; i < NF; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[(g * NF + i) & 7]));
#pragma unroll
        for (int i = 0; i < NC; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(f[(g * NC + i) & 7]));
#pragma unroll
        for (int i = 0; i < NS; ++i) asm volatile("s_add_i32 %0, %0, %1" : "+s"(s0) : "s"(s1));
#pragma unroll
        for (int i = 0; i < NN; ++i) asm volatile("s_nop 0");
#pragma unroll
        for (int i = 0; i < NW; ++i) asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
        for (int i = 0; i < NL; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[(g * NL + i) & 3]) : "v"(laddr), "n"(1024 * (i & 3)));
      }
    }
    if (NL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float sink = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) sink += acc[i][r];
  for (int i = 0; i < 8; ++i) sink += e[i] + f[i];
  for (int i = 0; i < 4; ++i) sink += (float)ld[i][0];
  if (threadIdx.x % 64 == 0) { out[threadIdx.x / 64] = t1 - t0; out[4 + threadIdx.x / 64] = (uint64_t)(sink + s0); }
}

template <int MF, int NE, int NF, int NS, int NL, int GRP = 1, int ACCA = 0, int NN = 0, int NW = 0, int NC = 0> static int run(uint64_t* d, const char* what) {
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MF, NE, NF, NS, NL, GRP, ACCA, NN, NW, NC>), dim3(1), dim3(256), 16384, 0, d, 0.001f, iters);
  HIP_OK(hipDeviceSynchronize());
  uint64_t h[8];
  HIP_OK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
  printf("%-44s %7.1f cycles per gap  (waves: %.1f %.1f %.1f %.1f)\n", what, h[0] / (4.0 * iters), h[0] / (4.0 * iters), h[1] / (4.0 * iters),
         h[2] / (4.0 * iters), h[3] / (4.0 * iters));
  return 0;
}

int main() {
  uint64_t* d; HIP_OK(hipMalloc(&d, 64));
  run<1, 0, 0, 0, 0>(d, "mfma only");
  run<0, 8, 0, 0, 0>(d, "8 exp (no mfma)");
  run<0, 0, 8, 0, 0>(d, "8 fma (no mfma)");
  run<0, 0, 0, 8, 0>(d, "8 salu (no mfma)");
  run<0, 0, 0, 0, 0, 1, 0, 0, 0, 8>(d, "8 cvt_pk (no mfma)");
  run<0, 0, 0, 0, 4>(d, "4 ds_read_b128, one wait per 16 (no mfma)");
  run<1, 0, 1, 0, 0>(d, "mfma + 1 fma");
  run<1, 0, 3, 0, 0>(d, "mfma + 3 fma");
  run<1, 0, 5, 0, 0>(d, "mfma + 5 fma");
  run<1, 0, 6, 0, 0>(d, "mfma + 6 fma");
  run<1, 0, 5, 0, 0, 2>(d, "2 mfma + 10 fma");
  run<1, 0, 5, 0, 0, 4>(d, "4 mfma + 20 fma");
  run<1, 0, 5, 0, 0, 1, 1>(d, "mfma(AGPR acc) + 5 fma");
  run<1, 0, 6, 0, 0, 1, 1>(d, "mfma(AGPR acc) + 6 fma");
  run<1, 0, 7, 0, 0, 1, 1>(d, "mfma(AGPR acc) + 7 fma");
  run<1, 2, 3, 0, 0, 1, 1>(d, "mfma(AGPR acc) + 2 exp + 3 fma");
  run<1, 1, 0, 0, 0, 1, 1>(d, "mfma(AGPR acc) + 1 exp");
  run<1, 2, 0, 0, 0, 1, 1>(d, "mfma(AGPR acc) + 2 exp");
  run<1, 3, 0, 0, 0, 1, 1>(d, "mfma(AGPR acc) + 3 exp");
  run<1, 0, 0, 5, 0>(d, "mfma + 5 salu");
  run<1, 0, 0, 7, 0>(d, "mfma + 7 salu");
  run<1, 0, 0, 9, 0>(d, "mfma + 9 salu");
  run<1, 0, 3, 3, 0>(d, "mfma + 3 fma + 3 salu");
  run<1, 0, 3, 4, 0>(d, "mfma + 3 fma + 4 salu");
  run<1, 0, 0, 0, 0, 1, 0, 5>(d, "mfma + 5 s_nop 0");
  run<1, 0, 0, 0, 0, 1, 0, 7>(d, "mfma + 7 s_nop 0");
  run<1, 0, 0, 0, 0, 1, 0, 0, 5>(d, "mfma + 5 s_waitcnt (nothing pending)");
  run<1, 0, 0, 0, 0, 1, 0, 0, 0, 5>(d, "mfma + 5 cvt_pk");
  run<1, 0, 0, 0, 1>(d, "mfma + 1 ds_read_b128");
  run<1, 0, 0, 0, 2>(d, "mfma + 2 ds_read_b128");
  run<1, 0, 3, 0, 2>(d, "mfma + 3 fma + 2 ds_read_b128");
  run<1, 2, 1, 1, 1>(d, "mfma + 2 exp + 1 fma + 1 salu + 1 ds_read");
  run<1, 1, 2, 1, 1>(d, "mfma + 1 exp + 2 fma + 1 salu + 1 ds_read");
  return 0;
}
