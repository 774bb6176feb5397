// DEV microbenchmark (round 5): do PACKED fp32 VALU ops (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: two lanes' worth of
// work per issue slot) cost one wave per SIMD less than the scalar forms beside its MFMAs?  The 64-row flash kernels are
// issue-bound (profiles/r04_ubench_issue.txt): every filler beyond ~5 per MFMA costs ~4.5 cycles.
// One workgroup of 4 waves (one per SIMD); per gap one v_mfma_f32_32x32x16_bf16 (AGPR accumulators) + the fillers named.
//   hipcc --offload-arch=gfx950 -O2 tools/r05/ubench_pk.hip -o gpurun_tools/ubench_pk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// per gap: NE v_exp_f32, NF v_fma_f32, NP v_pk_fma_f32, NA v_pk_add_f32, NM v_max3_f32, NC v_cvt_pk_bf16_f32, NL ds_read_b128
template <int NE, int NF, int NP, int NA, int NM, int NC, int NL>
__global__ __launch_bounds__(256, 1) void k(uint64_t* out, float seed, int iters) {
  extern __shared__ char lds[];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = seed * r;
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  float e[8], f[8], m[8];
  f32x2 p[8];
  for (int i = 0; i < 8; ++i) { e[i] = seed - i; f[i] = seed + i; m[i] = seed * i; p[i] = f32x2{seed + i, seed - i}; }
  u32x4 ld[4];
  const int laddr = (threadIdx.x & 63) * 16;
  for (int i = 0; i < 4; ++i) ld[i] = a;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[g]) : "v"(a), "v"(b));
#pragma unroll
      for (int i = 0; i < NE; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(e[(g * NE + i) & 7]));
#pragma unroll
      for (int i = 0; i < NF; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[(g * NF + i) & 7]));
#pragma unroll
      for (int i = 0; i < NP; ++i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[(g * NP + i) & 7]));
#pragma unroll
      for (int i = 0; i < NA; ++i) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[(g * NA + i + 4) & 7]));
#pragma unroll
      for (int i = 0; i < NM; ++i) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(m[(g * NM + i) & 7]));
#pragma unroll
      for (int i = 0; i < NC; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(f[(g * NC + i + 3) & 7]));
#pragma unroll
      for (int i = 0; i < NL; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[(g * NL + i) & 3]) : "v"(laddr), "n"(1024 * (i & 3)));
    }
    if (NL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float sink = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) sink += acc[i][r];
  for (int i = 0; i < 8; ++i) sink += e[i] + f[i] + m[i] + p[i][0] + p[i][1];
  for (int i = 0; i < 4; ++i) sink += (float)ld[i][0];
  if (threadIdx.x % 64 == 0) { out[threadIdx.x / 64] = t1 - t0; out[4 + threadIdx.x / 64] = (uint64_t)sink; }
}

template <int NE, int NF, int NP, int NA, int NM, int NC, int NL> static int run(uint64_t* d, const char* what) {
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NE, NF, NP, NA, NM, NC, NL>), dim3(1), dim3(256), 16384, 0, d, 0.001f, iters);
  HIP_OK(hipDeviceSynchronize());
  uint64_t h[8];
  HIP_OK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
  printf("%-64s %6.1f cycles per gap\n", what, h[0] / (4.0 * iters));
  return 0;
}

int main() {
  uint64_t* d; HIP_OK(hipMalloc(&d, 64));
  run<0, 0, 0, 0, 0, 0, 0>(d, "mfma only");
  run<0, 4, 0, 0, 0, 0, 0>(d, "mfma + 4 fma");
  run<0, 6, 0, 0, 0, 0, 0>(d, "mfma + 6 fma");
  run<0, 8, 0, 0, 0, 0, 0>(d, "mfma + 8 fma");
  run<0, 0, 2, 0, 0, 0, 0>(d, "mfma + 2 pk_fma   (= 4 fma of work)");
  run<0, 0, 3, 0, 0, 0, 0>(d, "mfma + 3 pk_fma   (= 6 fma)");
  run<0, 0, 4, 0, 0, 0, 0>(d, "mfma + 4 pk_fma   (= 8 fma)");
  run<0, 0, 6, 0, 0, 0, 0>(d, "mfma + 6 pk_fma   (= 12 fma)");
  run<0, 0, 0, 3, 0, 0, 0>(d, "mfma + 3 pk_add");
  run<0, 0, 0, 6, 0, 0, 0>(d, "mfma + 6 pk_add");
  // the forward's per-MFMA mix today: 1 fma + 1 exp + 1 add + 0.5 max3 + 0.5 cvt_pk (+ 0.5 LDS read); x2 = per two gaps
  run<1, 2, 0, 0, 0, 0, 0>(d, "fwd mix, scalar : 1 exp + 2 fma(fma,add)");
  run<1, 2, 0, 0, 1, 1, 0>(d, "fwd mix, scalar : 1 exp + 2 fma + 1 max3 + 1 cvt  (7.4/MFMA kernel ~)");
  run<1, 0, 1, 0, 1, 1, 0>(d, "fwd mix, packed : 1 exp + 1 pk_fma(=fma+..) + 1 max3 + 1 cvt");
  run<1, 0, 1, 1, 1, 1, 0>(d, "fwd mix, packed': 1 exp + 1 pk_fma + 1 pk_add + 1 max3 + 1 cvt");
  run<1, 2, 0, 0, 1, 1, 1>(d, "fwd mix, scalar + 1 ds_read_b128");
  run<1, 0, 1, 1, 1, 1, 1>(d, "fwd mix, packed' + 1 ds_read_b128");
  // dq tile: per MFMA 0.67 fma + 0.67 exp + 1.33 (sub, mul) + 0.33 cvt
  run<1, 3, 0, 0, 0, 1, 0>(d, "dq mix, scalar  : 1 exp + 3 fma + 1 cvt");
  run<1, 0, 1, 1, 0, 1, 0>(d, "dq mix, packed  : 1 exp + 1 pk_fma + 1 pk_add + 1 cvt");
  run<1, 0, 2, 0, 0, 1, 0>(d, "dq mix, packed' : 1 exp + 2 pk_fma + 1 cvt");
  return 0;
}
