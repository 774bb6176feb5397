// kbench -- native test/bench harness for libusp_hip.so (no python, no torch: starts in ms on a
// fresh GPU box).  TEST TOOL: links the CPU oracle (oracle/attn_oracle.c) as the checker.
//
//   kbench probe                         hardware-layout probes (MFMA operand/result maps, tr-read)
//   kbench fwd  B Sq Sk Hq Hkv D causal dtype check iters    forward: check vs oracle and/or time
//   kbench fwdmerge B Sq Sk Hq Hkv D dtype                    fused-merge path vs oracle (2 KV halves)
//   kbench bwd  B Sq Sk Hq Hkv D causal dtype check iters    backward
//   kbench pieces [Sq Sk H D iters]      merged launches of q rows x K/V row ranges (zigzag fetch waves): time per W
//   kbench split [S H iters]             few-head causal launch: one launch vs three concurrent launches cut along K
//   kbench ksplit B Sq Sk Hq Hkv D causal dtype check iters   forward with k_splits = 1, 2, 4 (ABI v4): check + time
//   kbench suite                          the standard correctness list + C2-shape timings
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "usp_hip.h"

extern "C" {
void usp_oracle_attn_fwd(const float* q, const float* k, const float* v, int B, int Sq, int Sk, int Hq,
                         int Hkv, int D, float scale, int causal, float* out, float* lse);
void usp_oracle_attn_bwd(const float* dout, const float* q, const float* k, const float* v,
                         const float* out, const float* lse, int B, int Sq, int Sk, int Hq, int Hkv,
                         int D, float scale, int causal, float* dq, float* dk, float* dv);
}

#define HIP_OK(x)                                                                       \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                          \
    }                                                                                   \
  } while (0)

// ---------------------------------------------------------------------------------------------
// host-side 16-bit helpers
// ---------------------------------------------------------------------------------------------
static uint16_t f2bf(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffff) > 0x7f800000) return 0x7fc0;
  u += 0x7fff + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
static uint16_t enc(float f, int dt) { return dt == 0 ? f2bf(f) : f2h(f); }
static float dec(uint16_t u, int dt) { return dt == 0 ? bf2f(u) : h2f(u); }

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 1) {}
  uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); }
  float uni() { return (next() + 0.5f) / 2147483648.0f; }
  float normal() { float u1 = uni(), u2 = uni(); return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }
};

// random N(0,1) tensor rounded to the 16-bit dtype: `bits` for the device, `vals` for the oracle
static void fill(std::vector<uint16_t>& bits, std::vector<float>& vals, size_t n, int dt, uint64_t seed,
                 float scale = 1.f) {
  bits.resize(n); vals.resize(n);
  Rng r(seed);
  for (size_t i = 0; i < n; ++i) { bits[i] = enc(r.normal() * scale, dt); vals[i] = dec(bits[i], dt); }
}

// Every uploaded tensor is followed by kPoisonTail bytes of 0xff (NaN in bf16, fp16 and fp32): a kernel whose bounds handling
// lets a lane read past the last valid row -- a descriptor clamp that does not cover the scalar offset of a piece, a ragged
// tail tile -- multiplies a NaN into its result instead of whatever the allocator left behind the buffer, and every CHECK
// of the suite (they all count NaN as bad) becomes a bounds test.
static const size_t kPoisonTail = 1 << 20;
template <typename T> static T* dev_upload(const std::vector<T>& h) {
  char* d; HIP_OK(hipMalloc(&d, h.size() * sizeof(T) + kPoisonTail));
  HIP_OK(hipMemset(d + h.size() * sizeof(T), 0xff, kPoisonTail));
  HIP_OK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return (T*)d;
}
template <typename T> static T* dev_alloc(size_t n, int fillbyte = 0xff) {
  T* d; HIP_OK(hipMalloc(&d, n * sizeof(T)));
  HIP_OK(hipMemset(d, fillbyte, n * sizeof(T)));
  return d;
}
template <typename T> static std::vector<T> dev_download(const T* d, size_t n) {
  std::vector<T> h(n);
  HIP_OK(hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost));
  return h;
}

// USP_KBENCH_ROWSTRIDE=<elements> (B = 1 only): the 16-bit inputs of `bwd` / `fwd` live in buffers whose rows are that far
// apart -- a head's rows then span more than 2^31 bytes with a few thousand rows (the span limits of the kernels' byte
// arithmetic are what this exercises).  Compact host data is scattered with hipMemcpy2D.
static int64_t env_rowstride() { const char* e = getenv("USP_KBENCH_ROWSTRIDE"); return e ? atoll(e) : 0; }
static uint16_t* dev_upload_rows(const std::vector<uint16_t>& h, int S, int row_elems) {
  const int64_t rs = env_rowstride();
  if (rs <= 0) return dev_upload(h);
  uint16_t* d; HIP_OK(hipMalloc(&d, (size_t)S * rs * 2));
  HIP_OK(hipMemset(d, 0x7f, (size_t)S * rs * 2));             // 0x7f7f = a large finite bf16 between the rows
  HIP_OK(hipMemcpy2D(d, (size_t)rs * 2, h.data(), (size_t)row_elems * 2, (size_t)row_elems * 2, S, hipMemcpyHostToDevice));
  return d;
}
static usp_tensor bshd_rows(void* p, int S, int H, int D) {
  const int64_t rs = env_rowstride();
  usp_tensor t; t.ptr = p; t.stride_s = rs > 0 ? rs : (int64_t)H * D; t.stride_b = (int64_t)S * t.stride_s; t.stride_h = D;
  return t;
}

static usp_tensor bshd(void* p, int S, int H, int D) {
  usp_tensor t; t.ptr = p; t.stride_b = (int64_t)S * H * D; t.stride_s = (int64_t)H * D; t.stride_h = D;
  return t;
}

struct Err { double max_abs = 0, max_rel_viol = 0; size_t bad = 0, n = 0, nan = 0; };
static Err compare(const float* got, const float* want, size_t n, double atol, double rtol) {
  Err e; e.n = n;
  for (size_t i = 0; i < n; ++i) {
    if (isinf(want[i]) && isinf(got[i]) && (want[i] < 0) == (got[i] < 0)) continue;
    if (got[i] != got[i]) { e.nan++; e.bad++; continue; }
    double d = fabs((double)got[i] - want[i]);
    if (d > e.max_abs) e.max_abs = d;
    if (d > atol + rtol * fabs(want[i])) e.bad++;
  }
  return e;
}

static double attn_flops(int B, int Sq, int Sk, int Hq, int D, int causal) {
  double pairs = 0;
  if (!causal) pairs = (double)Sq * Sk;
  else for (int i = 0; i < Sq; ++i) { long v = (long)i + (Sk - Sq) + 1; if (v > Sk) v = Sk; if (v > 0) pairs += v; }
  return 4.0 * B * Hq * pairs * D;
}

// Run `f` back-to-back for ~150 ms so the timing that follows sees the steady-state (DVFS-settled)
// clock: a 10 ms timing window right after process start reads ~10 % low.
template <typename F> static void warm_up(F f) {
  hipEvent_t a, b; HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b));
  float ms = 0;
  for (int round = 0; round < 50 && ms < 150.f; ++round) {
    HIP_OK(hipEventRecord(a, 0));
    for (int i = 0; i < 10; ++i) f();
    HIP_OK(hipEventRecord(b, 0)); HIP_OK(hipEventSynchronize(b));
    float t; HIP_OK(hipEventElapsedTime(&t, a, b)); ms += t;
  }
}

// ---------------------------------------------------------------------------------------------
// probes
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__global__ void probe_mfma(const uint16_t* A /*[32][16]*/, const uint16_t* Bm /*[16][32]*/, float* Dm /*[32][32]*/) {
  const int l = threadIdx.x, l31 = l & 31, hi = l >> 5;
  union { bf16x8 v; uint16_t u[8]; } a, b;
  for (int e = 0; e < 8; ++e) { a.u[e] = A[l31 * 16 + 8 * hi + e]; b.u[e] = Bm[(8 * hi + e) * 32 + l31]; }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) Dm[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31] = c[r];
}

__global__ void probe_tr(uint16_t* out /*[64][4]*/) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4 * 64];   // 4 groups x [4][16]
  const int l = threadIdx.x;
  for (int i = l; i < 256; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int grp = l >> 4, i = l & 15;
  // lane i of a group supplies chunk i of its [4][16] block: row i>>2, cols 4*(i&3)..
  const uint16_t* p = lds + grp * 64 + (i >> 2) * 16 + (i & 3) * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}

__global__ void probe_swap(uint32_t* out /*[64][2]*/) {
  const uint32_t u = threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  out[threadIdx.x * 2] = r[0];
  out[threadIdx.x * 2 + 1] = r[1];
}

static int run_probe() {
  int fails = 0;
  {  // MFMA layout, asymmetric small-integer operands (exact in bf16)
    std::vector<uint16_t> A(32 * 16), Bm(16 * 32);
    std::vector<float> Af(32 * 16), Bf(16 * 32);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) { float x = (float)((i * 3 + k * 5) % 7 - 3); A[i * 16 + k] = f2bf(x); Af[i * 16 + k] = x; }
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) { float x = (float)((k * 11 + j * 13) % 5 - 2); Bm[k * 32 + j] = f2bf(x); Bf[k * 32 + j] = x; }
    uint16_t* dA = dev_upload(A); uint16_t* dB = dev_upload(Bm); float* dD = dev_alloc<float>(32 * 32);
    hipLaunchKernelGGL(probe_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    HIP_OK(hipDeviceSynchronize());
    auto Dh = dev_download(dD, 32 * 32);
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      float ref = 0; for (int k = 0; k < 16; ++k) ref += Af[i * 16 + k] * Bf[k * 32 + j];
      if (Dh[i * 32 + j] != ref) { if (bad < 5) printf("  mfma mismatch D[%d][%d] = %g want %g\n", i, j, Dh[i * 32 + j], ref); bad++; }
    }
    printf("PROBE mfma_32x32x16_bf16 layout: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
    fails += bad != 0;
  }
  {  // ds_read_b64_tr_b16: lane i of a 16-group gets column i of the group's [4][16] block
    uint16_t* d = dev_alloc<uint16_t>(256);
    hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, 0, d);
    HIP_OK(hipDeviceSynchronize());
    auto h = dev_download(d, 256);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
      int want = (l >> 4) * 64 + j * 16 + (l & 15);
      if (h[l * 4 + j] != want) bad++;
    }
    printf("PROBE ds_read_b64_tr_b16 semantics: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
    if (bad) for (int l = 0; l < 64; ++l) printf("  lane %2d: %3d %3d %3d %3d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    fails += bad != 0;
  }
  {
    uint32_t* d = dev_alloc<uint32_t>(128);
    hipLaunchKernelGGL(probe_swap, dim3(1), dim3(64), 0, 0, d);
    HIP_OK(hipDeviceSynchronize());
    auto h = dev_download(d, 128);
    int bad = 0;
    for (int l = 0; l < 64; ++l) { if (h[2 * l] != (uint32_t)(l & 31)) bad++; if (h[2 * l + 1] != (uint32_t)((l & 31) + 32)) bad++; }
    printf("PROBE permlane32_swap(u,u) -> {[lo|lo],[hi|hi]}: %s\n", bad ? "FAIL" : "ok");
    if (bad) for (int l = 0; l < 64; l += 8) printf("  lane %2d: r0=%u r1=%u\n", l, h[2 * l], h[2 * l + 1]);
    fails += bad != 0;
  }
  hipDeviceProp_t prop; HIP_OK(hipGetDeviceProperties(&prop, 0));
  printf("DEVICE %s  CUs=%d  clock=%d MHz  LDS/block=%zu\n", prop.gcnArchName, prop.multiProcessorCount,
         prop.clockRate / 1000, prop.sharedMemPerBlock);
  return fails;
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
struct Tol { double out_atol, out_rtol, lse_atol; };
static Tol tol_for(int dt) { return dt == 0 ? Tol{2e-2, 2e-2, 2e-3} : Tol{4e-3, 4e-3, 1e-3}; }

// USP_KBENCH_FLAGS=1: time / check the launches with USP_LAUNCH_INTERLEAVE, as the ring schedules issue them
static int env_flags() { const char* e = getenv("USP_KBENCH_FLAGS"); return e ? atoi(e) : 0; }

static int run_fwd(int B, int Sq, int Sk, int Hq, int Hkv, int D, int causal, int dt, int check, int iters) {
  const size_t nq = (size_t)B * Sq * Hq * D, nk = (size_t)B * Sk * Hkv * D, nl = (size_t)B * Hq * Sq;
  std::vector<uint16_t> qb, kb, vb; std::vector<float> qf, kf, vf;
  fill(qb, qf, nq, dt, 1); fill(kb, kf, nk, dt, 2); fill(vb, vf, nk, dt, 3);
  uint16_t *dq = dev_upload(qb), *dk = dev_upload(kb), *dv = dev_upload(vb);
  uint16_t* dout = dev_alloc<uint16_t>(nq);
  float* dlse = dev_alloc<float>(nl);
  usp_fwd_args a; memset(&a, 0, sizeof(a)); a.flags = env_flags();
  a.dtype = dt; a.B = B; a.Sq = Sq; a.Sk = Sk; a.Hq = Hq; a.Hkv = Hkv; a.D = D; a.causal = causal;
  a.softmax_scale = 1.f / sqrtf((float)D);
  a.q = bshd(dq, Sq, Hq, D); a.k = bshd(dk, Sk, Hkv, D); a.v = bshd(dv, Sk, Hkv, D);
  a.out = bshd(dout, Sq, Hq, D);
  a.lse = dlse; a.lse_stride_b = (int64_t)Hq * Sq; a.lse_stride_h = Sq;
  a.merge_in = 0; a.final_begin = 0; a.final_end = Sq;
  int rc = usp_flash_fwd(&a, nullptr);
  if (rc) { printf("FWD launch failed: %s\n", usp_strerror(rc)); return 1; }
  HIP_OK(hipDeviceSynchronize());
  int fail = 0;
  char tag[160];
  snprintf(tag, sizeof tag, "fwd B%d Sq%d Sk%d Hq%d Hkv%d D%d %s %s", B, Sq, Sk, Hq, Hkv, D,
           causal ? "causal" : "full", dt ? "fp16" : "bf16");
  if (check) {
    std::vector<float> ro(nq), rl(nl);
    usp_oracle_attn_fwd(qf.data(), kf.data(), vf.data(), B, Sq, Sk, Hq, Hkv, D, a.softmax_scale, causal, ro.data(), rl.data());
    auto ob = dev_download(dout, nq); auto lh = dev_download(dlse, nl);
    std::vector<float> of(nq); for (size_t i = 0; i < nq; ++i) of[i] = dec(ob[i], dt);
    Tol t = tol_for(dt);
    Err eo = compare(of.data(), ro.data(), nq, t.out_atol, t.out_rtol);
    Err el = compare(lh.data(), rl.data(), nl, t.lse_atol, 1e-4);
    fail = (eo.bad || el.bad);
    printf("CHECK %-58s out max|err| %.3e bad %zu nan %zu | lse max|err| %.3e bad %zu  %s\n", tag, eo.max_abs,
           eo.bad, eo.nan, el.max_abs, el.bad, fail ? "FAIL" : "ok");
  }
  if (iters > 0) {
    hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    warm_up([&] { usp_flash_fwd(&a, nullptr); });
    HIP_OK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) usp_flash_fwd(&a, nullptr);
    HIP_OK(hipEventRecord(e1, 0)); HIP_OK(hipEventSynchronize(e1));
    float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    double fl = attn_flops(B, Sq, Sk, Hq, D, causal);
    printf("TIME  %-58s %.4f ms  %.1f TFLOP/s  (%.1f%% of 2500)\n", tag, ms, fl / ms * 1e-9, fl / ms * 1e-9 / 25.0);
  }
  hipFree(dq); hipFree(dk); hipFree(dv); hipFree(dout); hipFree(dlse);
  return fail;
}

// ---- overlap: can a transfer-like kernel on another stream run BESIDE the ring-step attention launches? -------
// The stand-in for RCCL's send/recv kernel is a copy with RCCL's footprint: a handful of workgroups that stay
// resident for the whole transfer (here: `wgs` workgroups move `mib` MiB; 8 workgroups reach roughly the
// bandwidth of one xGMI link).  Stream A: `steps` ring-step forward launches (C5 rank block, non-causal);
// stream B: steps-1 such copies back to back.  overlap = 1 - (t_both - t_attention) / t_copies.
__global__ void link_like_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

static int run_overlap(int steps, int mib, int wgs) {
  // default: the C5 ring-step block; USP_OVL_SHAPE="Sq Sk Hq Hkv" for another launch size (e.g. a head group's)
  int Sq = 16384, Sk = 8192, Hq = 16, Hkv = 2;
  if (const char* e = getenv("USP_OVL_SHAPE")) sscanf(e, "%d %d %d %d", &Sq, &Sk, &Hq, &Hkv);
  const int B = 1, D = 128, dt = 0;
  const size_t nq = (size_t)B * Sq * Hq * D, nk = (size_t)B * Sk * Hkv * D, nl = (size_t)B * Hq * Sq;
  std::vector<uint16_t> qb, kb, vb; std::vector<float> qf, kf, vf;
  fill(qb, qf, nq, dt, 1); fill(kb, kf, nk, dt, 2); fill(vb, vf, nk, dt, 3);
  uint16_t *dq = dev_upload(qb), *dk = dev_upload(kb), *dv = dev_upload(vb);
  uint16_t* dout = dev_alloc<uint16_t>(nq);
  float* dlse = dev_alloc<float>(nl);
  const size_t n16 = (size_t)mib * (1 << 20) / 16;
  uint4 *csrc = dev_alloc<uint4>(n16, 0x11), *cdst = dev_alloc<uint4>(n16, 0x22);
  usp_fwd_args a; memset(&a, 0, sizeof(a));
  a.dtype = dt; a.B = B; a.Sq = Sq; a.Sk = Sk; a.Hq = Hq; a.Hkv = Hkv; a.D = D; a.causal = 0;
  a.softmax_scale = 1.f / sqrtf((float)D);
  a.q = bshd(dq, Sq, Hq, D); a.k = bshd(dk, Sk, Hkv, D); a.v = bshd(dv, Sk, Hkv, D);
  a.out = bshd(dout, Sq, Hq, D);
  a.lse = dlse; a.lse_stride_b = (int64_t)Hq * Sq; a.lse_stride_h = Sq;
  a.final_begin = 0; a.final_end = Sq;
  hipStream_t sa, sb; HIP_OK(hipStreamCreate(&sa)); HIP_OK(hipStreamCreate(&sb));
  hipEvent_t e0, ea, eb; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&ea)); HIP_OK(hipEventCreate(&eb));
  auto attention = [&] { for (int i = 0; i < steps; ++i) usp_flash_fwd(&a, sa); };
  auto copies = [&] {
    for (int i = 0; i + 1 < steps; ++i)
      hipLaunchKernelGGL(link_like_copy, dim3(wgs), dim3(256), 0, sb, csrc, cdst, n16);
  };
  auto measure = [&](bool do_a, bool do_b) {          // wall time from a common start to both streams idle
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      HIP_OK(hipDeviceSynchronize());
      HIP_OK(hipEventRecord(e0, sa));
      HIP_OK(hipStreamWaitEvent(sb, e0, 0));
      if (do_b) copies();                             // queued first, as KVRelay queues the transfers up-front
      if (do_a) attention();
      HIP_OK(hipEventRecord(ea, sa)); HIP_OK(hipEventRecord(eb, sb));
      HIP_OK(hipEventSynchronize(ea)); HIP_OK(hipEventSynchronize(eb));
      float ta, tb; HIP_OK(hipEventElapsedTime(&ta, e0, ea)); HIP_OK(hipEventElapsedTime(&tb, e0, eb));
      const float t = ta > tb ? ta : tb;
      best = t < best ? t : best;
    }
    return best;
  };
  warm_up([&] { usp_flash_fwd(&a, sa); });
  const float t_copy = measure(false, true);
  printf("OVERLAP copies alone: %d x %d MiB on %d workgroups: %.3f ms (%.0f GB/s each way)\n", steps - 1, mib, wgs,
         t_copy, (steps - 1) * (double)mib * 1.048576 / t_copy);
  for (int inter = 0; inter < 2; ++inter) {
    a.flags = inter ? USP_LAUNCH_INTERLEAVE : 0;
    const float t_att = measure(true, false), t_both = measure(true, true);
    printf("OVERLAP %-26s attention alone %.3f ms | with copies %.3f ms | overlap = %.3f\n",
           inter ? "USP_LAUNCH_INTERLEAVE" : "persistent launches", t_att, t_both, 1.f - (t_both - t_att) / t_copy);
  }
  hipFree(dq); hipFree(dk); hipFree(dv); hipFree(dout); hipFree(dlse); hipFree(csrc); hipFree(cdst);
  return 0;
}

// fused merge: KV split in two halves, two calls (acc write, then merge_in + final) == one full call.
// Also exercises the row-range logic: rows [0, Sq/2) are declared final already in call 1 when
// `partial_final` is set (they then must NOT be touched by call 2, which only covers rows Sq/2..).
static int run_fwdmerge(int B, int Sq, int Sk, int Hq, int Hkv, int D, int dt) {
  const size_t nq = (size_t)B * Sq * Hq * D, nk = (size_t)B * Sk * Hkv * D, nl = (size_t)B * Hq * Sq;
  std::vector<uint16_t> qb, kb, vb; std::vector<float> qf, kf, vf;
  fill(qb, qf, nq, dt, 11); fill(kb, kf, nk, dt, 12); fill(vb, vf, nk, dt, 13);
  uint16_t *dq = dev_upload(qb), *dk = dev_upload(kb), *dv = dev_upload(vb);
  uint16_t* dout = dev_alloc<uint16_t>(nq);
  float* dacc = dev_alloc<float>(nq);
  float* dlse = dev_alloc<float>(nl);
  const int h1 = (Sk / 2 + 7) / 8 * 8 > 0 ? (Sk / 2) : 1;
  usp_fwd_args a; memset(&a, 0, sizeof(a));
  a.dtype = dt; a.B = B; a.Sq = Sq; a.Hq = Hq; a.Hkv = Hkv; a.D = D; a.causal = 0;
  a.softmax_scale = 1.f / sqrtf((float)D);
  a.q = bshd(dq, Sq, Hq, D); a.out = bshd(dout, Sq, Hq, D); a.acc = bshd(dacc, Sq, Hq, D);
  a.lse = dlse; a.lse_stride_b = (int64_t)Hq * Sq; a.lse_stride_h = Sq;
  // call 1: keys [0,h1), write acc
  a.Sk = h1; a.k = bshd(dk, Sk, Hkv, D); a.v = bshd(dv, Sk, Hkv, D);
  a.merge_in = 0; a.final_begin = 0; a.final_end = 0;
  int rc = usp_flash_fwd(&a, nullptr);
  // call 2: keys [h1,Sk), merge + final
  a.Sk = Sk - h1;
  a.k.ptr = dk + (size_t)h1 * Hkv * D; a.v.ptr = dv + (size_t)h1 * Hkv * D;
  a.merge_in = 1; a.final_begin = 0; a.final_end = Sq;
  rc |= usp_flash_fwd(&a, nullptr);
  if (rc) { printf("FWDMERGE launch failed: %s\n", usp_strerror(rc)); return 1; }
  HIP_OK(hipDeviceSynchronize());
  std::vector<float> ro(nq), rl(nl);
  usp_oracle_attn_fwd(qf.data(), kf.data(), vf.data(), B, Sq, Sk, Hq, Hkv, D, a.softmax_scale, 0, ro.data(), rl.data());
  auto ob = dev_download(dout, nq); auto lh = dev_download(dlse, nl);
  std::vector<float> of(nq); for (size_t i = 0; i < nq; ++i) of[i] = dec(ob[i], dt);
  Tol t = tol_for(dt);
  Err eo = compare(of.data(), ro.data(), nq, t.out_atol, t.out_rtol);
  Err el = compare(lh.data(), rl.data(), nl, t.lse_atol, 1e-4);
  int fail = (eo.bad || el.bad);
  printf("CHECK fwdmerge B%d Sq%d Sk%d(%d+%d) Hq%d Hkv%d D%d %s : out max|err| %.3e bad %zu | lse max|err| %.3e bad %zu  %s\n",
         B, Sq, Sk, h1, Sk - h1, Hq, Hkv, D, dt ? "fp16" : "bf16", eo.max_abs, eo.bad, el.max_abs, el.bad, fail ? "FAIL" : "ok");
  hipFree(dq); hipFree(dk); hipFree(dv); hipFree(dout); hipFree(dacc); hipFree(dlse);
  return fail;
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
static int run_bwd(int B, int Sq, int Sk, int Hq, int Hkv, int D, int causal, int dt, int check, int iters) {
  const size_t nq = (size_t)B * Sq * Hq * D, nk = (size_t)B * Sk * Hkv * D, nl = (size_t)B * Hq * Sq;
  std::vector<uint16_t> qb, kb, vb, dob; std::vector<float> qf, kf, vf, dof;
  fill(qb, qf, nq, dt, 21); fill(kb, kf, nk, dt, 22); fill(vb, vf, nk, dt, 23); fill(dob, dof, nq, dt, 24);
  uint16_t *dq_ = dev_upload_rows(qb, Sq, Hq * D), *dk_ = dev_upload_rows(kb, Sk, Hkv * D), *dv_ = dev_upload_rows(vb, Sk, Hkv * D),
           *ddo = dev_upload_rows(dob, Sq, Hq * D);
  uint16_t* dout = dev_alloc<uint16_t>(nq);
  float* dlse = dev_alloc<float>(nl);
  float* ddelta = dev_alloc<float>(nl);
  float *gdq = dev_alloc<float>(nq), *gdk = dev_alloc<float>(nk), *gdv = dev_alloc<float>(nk);
  const float scale = 1.f / sqrtf((float)D);
  usp_fwd_args f; memset(&f, 0, sizeof(f));
  f.dtype = dt; f.B = B; f.Sq = Sq; f.Sk = Sk; f.Hq = Hq; f.Hkv = Hkv; f.D = D; f.causal = causal;
  f.softmax_scale = scale;
  f.q = bshd_rows(dq_, Sq, Hq, D); f.k = bshd_rows(dk_, Sk, Hkv, D); f.v = bshd_rows(dv_, Sk, Hkv, D); f.out = bshd(dout, Sq, Hq, D);
  f.lse = dlse; f.lse_stride_b = (int64_t)Hq * Sq; f.lse_stride_h = Sq; f.final_end = Sq;
  int rc = usp_flash_fwd(&f, nullptr);
  usp_tensor tdo = bshd_rows(ddo, Sq, Hq, D), tout = bshd(dout, Sq, Hq, D);
  rc |= usp_bwd_delta(dt, B, Sq, Hq, D, &tdo, &tout, ddelta, (int64_t)Hq * Sq, Sq, nullptr);
  usp_bwd_args a; memset(&a, 0, sizeof(a)); a.flags = env_flags();
  a.dtype = dt; a.B = B; a.Sq = Sq; a.Sk = Sk; a.Hq = Hq; a.Hkv = Hkv; a.D = D; a.causal = causal;
  a.softmax_scale = scale;
  a.dout = tdo; a.q = f.q; a.k = f.k; a.v = f.v;
  a.lse = dlse; a.delta = ddelta;
  a.lse_stride_b = a.delta_stride_b = (int64_t)Hq * Sq; a.lse_stride_h = a.delta_stride_h = Sq;
  a.dq = bshd(gdq, Sq, Hq, D); a.dk = bshd(gdk, Sk, Hkv, D); a.dv = bshd(gdv, Sk, Hkv, D);
  if (const char* e = getenv("USP_KBENCH_BWD_SPLITS")) {           // "dq,dkdv": cuts of the two launches (ABI v5)
    int x = 0, y = 0;
    if (sscanf(e, "%d,%d", &x, &y) == 2) { a.dq_splits = x; a.dkdv_splits = y; }
  }
  if (const char* e = getenv("USP_KBENCH_BWD_HEADS")) {            // query heads per dK/dV work item (ABI v7); not a divisor: the library's choice
    const int g = atoi(e);
    if (g > 0 && (Hq / Hkv) % g == 0) a.dkdv_heads = g;
  }
  const int64_t wsb = getenv("USP_NO_WORKSPACE") ? 0 : usp_flash_bwd_workspace_bytes(&a);
  if (wsb > 0) { a.workspace = dev_alloc<char>((size_t)wsb); a.workspace_bytes = wsb; }
  rc |= usp_flash_bwd(&a, nullptr);
  if (rc) { printf("BWD launch failed: %s\n", usp_strerror(rc)); return 1; }
  HIP_OK(hipDeviceSynchronize());
  char tag[160];
  char hd[24] = "";
  if (a.dkdv_heads) snprintf(hd, sizeof hd, " heads/item %d", a.dkdv_heads);
  snprintf(tag, sizeof tag, "bwd B%d Sq%d Sk%d Hq%d Hkv%d D%d %s %s%s%s%s", B, Sq, Sk, Hq, Hkv, D,
           causal ? "causal" : "full", dt ? "fp16" : "bf16", getenv("USP_KBENCH_BWD_SPLITS") ? " cuts " : "",
           getenv("USP_KBENCH_BWD_SPLITS") ? getenv("USP_KBENCH_BWD_SPLITS") : "", hd);
  int fail = 0;
  if (check) {
    std::vector<float> ro(nq), rl(nl), rdq(nq), rdk(nk), rdv(nk);
    usp_oracle_attn_fwd(qf.data(), kf.data(), vf.data(), B, Sq, Sk, Hq, Hkv, D, scale, causal, ro.data(), rl.data());
    usp_oracle_attn_bwd(dof.data(), qf.data(), kf.data(), vf.data(), ro.data(), rl.data(), B, Sq, Sk, Hq, Hkv, D,
                        scale, causal, rdq.data(), rdk.data(), rdv.data());
    auto hq = dev_download(gdq, nq); auto hk = dev_download(gdk, nk); auto hv = dev_download(gdv, nk);
    const double at = dt == 0 ? 5e-2 : 1e-2;
    Err e1 = compare(hq.data(), rdq.data(), nq, at, at), e2 = compare(hk.data(), rdk.data(), nk, at, at),
        e3 = compare(hv.data(), rdv.data(), nk, at, at);
    fail = e1.bad || e2.bad || e3.bad;
    printf("CHECK %-58s dq %.3e (%zu bad) dk %.3e (%zu bad) dv %.3e (%zu bad)  %s\n", tag, e1.max_abs, e1.bad,
           e2.max_abs, e2.bad, e3.max_abs, e3.bad, fail ? "FAIL" : "ok");
  }
  if (iters > 0) {
    hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    warm_up([&] { usp_flash_bwd(&a, nullptr); });
    HIP_OK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) usp_flash_bwd(&a, nullptr);
    HIP_OK(hipEventRecord(e1, 0)); HIP_OK(hipEventSynchronize(e1));
    float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    double fl = 2.5 * attn_flops(B, Sq, Sk, Hq, D, causal);
    printf("TIME  %-58s %.4f ms  %.1f TFLOP/s  (%.1f%% of 2500)\n", tag, ms, fl / ms * 1e-9, fl / ms * 1e-9 / 25.0);
  }
  hipFree(dq_); hipFree(dk_); hipFree(dv_); hipFree(ddo); hipFree(dout); hipFree(dlse); hipFree(ddelta);
  hipFree(gdq); hipFree(gdk); hipFree(gdv);
  return fail;
}

static int run_ksplit(int B, int Sq, int Sk, int Hq, int Hkv, int D, int causal, int dt, int check, int iters);
static int suite(bool with_bwd) {
  int f = 0;
  f += run_probe();
  // correctness: tile-aligned, ragged, GQA, Sq != Sk (bottom-right causal), all head dims, both dtypes
  struct C { int B, Sq, Sk, Hq, Hkv, D, causal, dt; };
  const C cs[] = {
      {1, 256, 256, 1, 1, 128, 0, 0}, {1, 256, 256, 1, 1, 128, 1, 0}, {2, 512, 512, 4, 4, 128, 1, 0},
      {1, 1024, 1024, 8, 8, 64, 1, 0}, {1, 1024, 1024, 8, 8, 64, 1, 1}, {1, 384, 640, 4, 2, 128, 0, 0},
      {1, 320, 320, 4, 1, 128, 1, 1}, {2, 77, 77, 2, 2, 64, 1, 0}, {1, 200, 333, 3, 1, 128, 1, 0},
      {1, 333, 200, 2, 2, 128, 1, 0}, {1, 1, 1, 1, 1, 128, 1, 0}, {1, 65, 191, 2, 1, 32, 0, 1},
      {1, 512, 256, 2, 2, 128, 0, 0}, {1, 256, 512, 2, 2, 128, 0, 0}, {1, 2048, 2048, 2, 1, 128, 1, 0},
  };
  for (const C& c : cs) f += run_fwd(c.B, c.Sq, c.Sk, c.Hq, c.Hkv, c.D, c.causal, c.dt, 1, 0);
  f += run_fwdmerge(1, 256, 512, 2, 2, 128, 0);
  f += run_fwdmerge(2, 300, 200, 4, 2, 64, 1);
  f += run_fwdmerge(1, 64, 192, 2, 1, 128, 0);
  // K split (ABI v4): k_splits 1..8 x {plain, partly final, merge_in}; causal incl. Sq != Sk, GQA, ragged, every head dim
  f += run_ksplit(1, 1024, 1024, 2, 2, 128, 1, 0, 1, 0);
  f += run_ksplit(2, 300, 712, 4, 2, 64, 0, 1, 1, 0);
  f += run_ksplit(1, 333, 200, 2, 1, 128, 1, 0, 1, 0);
  f += run_ksplit(1, 200, 333, 3, 3, 32, 1, 1, 1, 0);
  f += run_ksplit(2, 77, 77, 2, 2, 64, 1, 0, 1, 0);
  if (with_bwd) {
    const C bs[] = {{1, 256, 256, 1, 1, 128, 0, 0}, {1, 256, 256, 2, 2, 128, 1, 0}, {2, 512, 512, 4, 2, 128, 1, 0},
                    {1, 384, 640, 4, 2, 64, 0, 1},  {1, 200, 333, 3, 1, 128, 1, 0}, {1, 333, 200, 2, 2, 64, 1, 0},
                    {2, 77, 77, 2, 2, 32, 1, 0},    {1, 1024, 1024, 8, 8, 64, 1, 0}};
    for (const C& c : bs) f += run_bwd(c.B, c.Sq, c.Sk, c.Hq, c.Hkv, c.D, c.causal, c.dt, 1, 0);
    // cuts of few-item launches (ABI v5): every shape again with the dQ launch cut along K and the dK/dV launch along Q
    for (const char* cuts : {"2,2", "3,4", "8,8", "4,1", "1,3"}) {
      setenv("USP_KBENCH_BWD_SPLITS", cuts, 1);
      for (const C& c : bs) f += run_bwd(c.B, c.Sq, c.Sk, c.Hq, c.Hkv, c.D, c.causal, c.dt, 1, 0);
    }
    unsetenv("USP_KBENCH_BWD_SPLITS");
    // the GQA loop inside a dK/dV work item (ABI v7 dkdv_heads): 1, 2, 4 and all heads of a group per item, with and without
    // cuts of the query rows on top; ragged, Sq != Sk both ways, D < 128 (the 8-wave kernel takes the same field)
    const C gs[] = {{1, 512, 512, 8, 2, 128, 1, 0}, {2, 333, 200, 8, 1, 128, 1, 0}, {1, 200, 333, 4, 1, 128, 0, 1},
                    {1, 1024, 1024, 8, 1, 128, 1, 1}, {1, 384, 640, 8, 2, 64, 1, 0}, {2, 130, 130, 6, 2, 128, 1, 0}};
    for (const char* heads : {"1", "2", "4", "8", "3", "6"}) {
      setenv("USP_KBENCH_BWD_HEADS", heads, 1);
      for (const C& c : gs)
        if ((c.Hq / c.Hkv) % atoi(heads) == 0) f += run_bwd(c.B, c.Sq, c.Sk, c.Hq, c.Hkv, c.D, c.causal, c.dt, 1, 0);
    }
    setenv("USP_KBENCH_BWD_SPLITS", "2,3", 1);
    for (const char* heads : {"2", "4"}) {
      setenv("USP_KBENCH_BWD_HEADS", heads, 1);
      for (const C& c : gs)
        if ((c.Hq / c.Hkv) % atoi(heads) == 0) f += run_bwd(c.B, c.Sq, c.Sk, c.Hq, c.Hkv, c.D, c.causal, c.dt, 1, 0);
    }
    unsetenv("USP_KBENCH_BWD_SPLITS");
    unsetenv("USP_KBENCH_BWD_HEADS");
  }
  printf("SUITE %s (%d failing groups)\n", f ? "FAIL" : "PASS", f);
  // timings at BASELINE shapes (C2 = B2 S8192 H16 D128 bf16 causal) and the C5 per-rank ring blocks
  run_fwd(2, 8192, 8192, 16, 16, 128, 1, 0, 0, 20);
  run_fwd(2, 8192, 8192, 16, 16, 128, 0, 0, 0, 10);
  run_fwd(1, 16384, 16384, 16, 2, 128, 1, 0, 0, 5);
  run_fwd(1, 16384, 8192, 16, 2, 128, 0, 0, 0, 5);
  run_fwd(1, 8192, 16384, 16, 2, 128, 0, 0, 0, 5);
  run_ksplit(1, 16384, 16384, 4, 4, 128, 1, 0, 0, 5);      // few-head launches: the head groups of the 2-GPU config
  run_ksplit(1, 16384, 16384, 2, 2, 128, 1, 0, 0, 5);
  if (with_bwd) {
    run_bwd(2, 8192, 8192, 16, 16, 128, 1, 0, 0, 5);
    run_bwd(1, 16384, 16384, 16, 2, 128, 1, 0, 0, 3);
  }
  return f;
}

// What the row-range waves of the zigzag mesh fetch cost in kernel time: the launches ring rank 0 issues behind step 0
// at ring degree P (every step reads both K/V halves of its peer with q[c:]: Sq = c query rows against P-1 peers x 2
// halves of Sk = c keys), each half cut into W row ranges -> (P-1) * 2W merged launches of Sq x Sk/W.  W = 1 is the
// two-wave fetch; the difference to W = 2, 4 is the price of the extra merge epilogues (tools/link_model.py charges
// 16 us per launch at configs[3]).  Interleavable launches (what runs beside a transfer).
static int run_pieces(int Sq, int Sk, int H, int D, int iters) {
  const int P = 4, dt = 0, B = 1;
  const size_t nq = (size_t)B * Sq * H * D, nk = (size_t)B * Sk * H * D, nl = (size_t)B * H * Sq;
  std::vector<uint16_t> qb, kb, vb; std::vector<float> qf, kf, vf;
  fill(qb, qf, nq, dt, 21); fill(kb, kf, nk, dt, 22); fill(vb, vf, nk, dt, 23);
  uint16_t *dq = dev_upload(qb), *dk = dev_upload(kb), *dv = dev_upload(vb);
  uint16_t* dout = dev_alloc<uint16_t>(nq);
  float* dacc = dev_alloc<float>(nq);
  float* dlse = dev_alloc<float>(nl);
  usp_fwd_args a; memset(&a, 0, sizeof(a));
  a.flags = USP_LAUNCH_INTERLEAVE;
  a.dtype = dt; a.B = B; a.Sq = Sq; a.Hq = H; a.Hkv = H; a.D = D; a.causal = 0;
  a.softmax_scale = 1.f / sqrtf((float)D);
  a.q = bshd(dq, Sq, H, D); a.out = bshd(dout, Sq, H, D); a.acc = bshd(dacc, Sq, H, D);
  a.lse = dlse; a.lse_stride_b = (int64_t)H * Sq; a.lse_stride_h = Sq;
  a.k = bshd(dk, Sk, H, D); a.v = bshd(dv, Sk, H, D);
  // something to merge into
  a.Sk = Sk; a.merge_in = 0; a.final_begin = 0; a.final_end = 0;
  if (int rc = usp_flash_fwd(&a, nullptr)) { printf("PIECES launch failed: %s\n", usp_strerror(rc)); return 1; }
  a.merge_in = 1;
  hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  float base = 0.f;
  for (int W : {1, 2, 4, 8}) {
    auto pass = [&] {
      for (int peer = 0; peer < P - 1; ++peer)
        for (int w = 0; w < 2 * W; ++w) {
          const int lo = (int)((int64_t)(w % W) * Sk / W), hi = (int)((int64_t)(w % W + 1) * Sk / W);
          a.Sk = hi - lo;
          a.k.ptr = dk + (size_t)lo * H * D; a.v.ptr = dv + (size_t)lo * H * D;
          usp_flash_fwd(&a, nullptr);
        }
    };
    warm_up(pass);
    HIP_OK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) pass();
    HIP_OK(hipEventRecord(e1, 0)); HIP_OK(hipEventSynchronize(e1));
    float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    if (W == 1) base = ms;
    const int n = (P - 1) * 2 * W;
    printf("PIECES Sq%d Sk%d H%d D%d  W=%d: %2d launches of %d x %d  %.4f ms  (+%.1f us per extra launch)  %.0f TFLOP/s\n", Sq, Sk, H, D, W, n,
           Sq, Sk / W, ms, W > 1 ? (ms - base) * 1e3 / (n - (P - 1) * 2) : 0.0,
           4.0 * B * H * (double)Sq * Sk * D * (P - 1) * 2 / (ms * 1e-3) / 1e12);
  }
  hipFree(dq); hipFree(dk); hipFree(dv); hipFree(dout); hipFree(dacc); hipFree(dlse);
  return 0;
}

// Few-head causal launches (the head groups of the 2-GPU config: B1 S16384, 2 or 4 heads) last as long as their heaviest
// item, whatever the placement.  Potential of cutting the work along K WITHOUT a new kernel: three launches on three
// streams -- rows [0,S/2) causal on keys [0,S/2); rows [S/2,S) full on keys [0,S/2); rows [S/2,S) causal on keys
// [S/2,S) -- the last two into separate fp32 partials (a merge of S/2 rows would follow; not timed: HBM-bound, ~10 us).
// The heaviest item then holds S/2 keys instead of S.  Timed: the single launch, and the three together.
static int run_split(int S, int H, int iters) {
  const int B = 1, D = 128, dt = 0, h = S / 2;
  const size_t nq = (size_t)B * S * H * D, nl = (size_t)B * H * S;
  std::vector<uint16_t> qb, kb, vb; std::vector<float> qf, kf, vf;
  fill(qb, qf, nq, dt, 31); fill(kb, kf, nq, dt, 32); fill(vb, vf, nq, dt, 33);
  uint16_t *dq = dev_upload(qb), *dk = dev_upload(kb), *dv = dev_upload(vb);
  uint16_t* dout = dev_alloc<uint16_t>(nq);
  float *acc1 = dev_alloc<float>(nq), *acc2 = dev_alloc<float>(nq);
  float *lse = dev_alloc<float>(nl), *lse1 = dev_alloc<float>(nl), *lse2 = dev_alloc<float>(nl);
  auto base = [&](int Sq, int Sk, int causal, uint16_t* q, uint16_t* k, uint16_t* v, float* l) {
    usp_fwd_args a; memset(&a, 0, sizeof(a));
    a.dtype = dt; a.B = B; a.Sq = Sq; a.Sk = Sk; a.Hq = H; a.Hkv = H; a.D = D; a.causal = causal;
    a.softmax_scale = 1.f / sqrtf((float)D);
    a.q = bshd(q, Sq, H, D); a.k = bshd(k, Sk, H, D); a.v = bshd(v, Sk, H, D);
    a.lse = l; a.lse_stride_b = (int64_t)H * Sq; a.lse_stride_h = Sq;
    return a;
  };
  const size_t half = (size_t)h * H * D;
  hipStream_t st[3]; for (auto& x : st) HIP_OK(hipStreamCreate(&x));
  hipEvent_t e0, e1[3]; HIP_OK(hipEventCreate(&e0)); for (auto& x : e1) HIP_OK(hipEventCreate(&x));
  for (int inter = 0; inter < 2; ++inter) {
    usp_fwd_args whole = base(S, S, 1, dq, dk, dv, lse);
    whole.out = bshd(dout, S, H, D); whole.final_end = S; whole.flags = inter ? USP_LAUNCH_INTERLEAVE : 0;
    usp_fwd_args a1 = base(h, h, 1, dq, dk, dv, lse);                       // rows [0,h) x keys [0,h), causal, final
    a1.out = bshd(dout, h, H, D); a1.final_end = h; a1.flags = whole.flags;
    usp_fwd_args a2 = base(h, h, 0, dq + half, dk, dv, lse1);               // rows [h,S) x keys [0,h), full -> partial 1
    a2.acc = bshd(acc1, h, H, D); a2.final_end = 0; a2.flags = whole.flags;
    usp_fwd_args a3 = base(h, h, 1, dq + half, dk + half, dv + half, lse2); // rows [h,S) x keys [h,S), causal -> partial 2
    a3.acc = bshd(acc2, h, H, D); a3.final_end = 0; a3.flags = whole.flags;
    auto time = [&](bool split) {
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipEventRecord(e0, st[0]));
        HIP_OK(hipStreamWaitEvent(st[1], e0, 0)); HIP_OK(hipStreamWaitEvent(st[2], e0, 0));
        for (int i = 0; i < iters; ++i) {
          if (!split) { usp_flash_fwd(&whole, st[0]); continue; }
          usp_flash_fwd(&a2, st[0]); usp_flash_fwd(&a3, st[1]); usp_flash_fwd(&a1, st[2]);   // heaviest first
        }
        float t = 0.f;
        for (int j = 0; j < 3; ++j) { HIP_OK(hipEventRecord(e1[j], st[j])); }
        for (int j = 0; j < 3; ++j) { HIP_OK(hipEventSynchronize(e1[j])); float x; HIP_OK(hipEventElapsedTime(&x, e0, e1[j])); t = x > t ? x : t; }
        t /= iters;
        best = t < best ? t : best;
      }
      return best;
    };
    if (int rc = usp_flash_fwd(&whole, st[0])) { printf("SPLIT launch failed: %s\n", usp_strerror(rc)); return 1; }
    if (int rc = usp_flash_fwd(&a1, st[0]) | usp_flash_fwd(&a2, st[0]) | usp_flash_fwd(&a3, st[0])) { printf("SPLIT launch failed: %s\n", usp_strerror(rc)); return 1; }
    const float t1 = time(false), t3 = time(true);
    const double fl = attn_flops(B, S, S, H, D, 1);
    printf("SPLIT B1 S%d H%d D128 causal %-12s one launch %.4f ms (%.0f TFLOP/s) | three launches cut along K, 3 streams %.4f ms (%.0f TFLOP/s)\n",
           S, H, inter ? "interleave" : "persistent", t1, fl / (t1 * 1e-3) / 1e12, t3, fl / (t3 * 1e-3) / 1e12);
  }
  hipFree(dq); hipFree(dk); hipFree(dv); hipFree(dout); hipFree(acc1); hipFree(acc2); hipFree(lse); hipFree(lse1); hipFree(lse2);
  return 0;
}

// K split (ABI v4): the same call with k_splits = 1 (off), 2, 3, 4 and 8 -- against the oracle (check != 0), in three
// forms: a plain final call, a call whose rows [Sq/2, Sq) are not final (fp32 acc), and a merge_in call on top of a
// first half of the keys (what a ring step does); and timed (iters > 0).
static int run_ksplit(int B, int Sq, int Sk, int Hq, int Hkv, int D, int causal, int dt, int check, int iters) {
  const size_t nq = (size_t)B * Sq * Hq * D, nk = (size_t)B * Sk * Hkv * D, nl = (size_t)B * Hq * Sq;
  std::vector<uint16_t> qb, kb, vb; std::vector<float> qf, kf, vf;
  fill(qb, qf, nq, dt, 41); fill(kb, kf, nk, dt, 42); fill(vb, vf, nk, dt, 43);
  uint16_t *dq = dev_upload(qb), *dk = dev_upload(kb), *dv = dev_upload(vb);
  uint16_t* dout = dev_alloc<uint16_t>(nq);
  float* dacc = dev_alloc<float>(nq);
  float* dlse = dev_alloc<float>(nl);
  usp_fwd_args a; memset(&a, 0, sizeof(a)); a.flags = env_flags();
  a.dtype = dt; a.B = B; a.Sq = Sq; a.Sk = Sk; a.Hq = Hq; a.Hkv = Hkv; a.D = D; a.causal = causal;
  a.softmax_scale = 1.f / sqrtf((float)D);
  a.q = bshd(dq, Sq, Hq, D); a.k = bshd(dk, Sk, Hkv, D); a.v = bshd(dv, Sk, Hkv, D);
  a.out = bshd(dout, Sq, Hq, D); a.acc = bshd(dacc, Sq, Hq, D);
  a.lse = dlse; a.lse_stride_b = (int64_t)Hq * Sq; a.lse_stride_h = Sq;
  const int64_t ws_bytes = usp_flash_fwd_workspace_bytes(&a, 8);
  void* ws = nullptr; HIP_OK(hipMalloc(&ws, ws_bytes));
  std::vector<float> ro(nq), rl(nl);
  if (check) usp_oracle_attn_fwd(qf.data(), kf.data(), vf.data(), B, Sq, Sk, Hq, Hkv, D, a.softmax_scale, causal, ro.data(), rl.data());
  Tol t = tol_for(dt);
  int fail = 0;
  hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  for (int n : {1, 2, 3, 4, 8}) {
    a.k_splits = n; a.workspace = n > 1 ? ws : nullptr;
    for (int form = 0; form < (check ? 3 : 1); ++form) {
      HIP_OK(hipMemset(dout, 0xff, nq * 2)); HIP_OK(hipMemset(dacc, 0xff, nq * 4)); HIP_OK(hipMemset(dlse, 0xff, nl * 4));
      a.merge_in = 0; a.final_begin = 0; a.final_end = Sq; a.Sk = Sk; a.k.ptr = dk; a.v.ptr = dv; a.causal = causal;
      int rc = 0;
      if (form == 1) a.final_end = Sq / 2;                       // rows [Sq/2, Sq) stay fp32 in acc
      if (form == 2) {                                           // keys [0,h) first (plain, to acc), then the rest merged in
        // causal alignment is bottom-right, so the first call must be the FULL block over the first keys of a
        // non-causal problem; with causal inputs this form is only run for Sq == Sk on the second half being causal-aligned
        const int h = (Sk / 2) & ~7;
        if (causal || h == 0) continue;
        usp_fwd_args f = a; f.k_splits = 0; f.workspace = nullptr; f.Sk = h; f.final_end = 0;
        rc = usp_flash_fwd(&f, nullptr);
        a.Sk = Sk - h; a.k.ptr = dk + (size_t)h * Hkv * D; a.v.ptr = dv + (size_t)h * Hkv * D; a.merge_in = 1;
      }
      rc |= usp_flash_fwd(&a, nullptr);
      if (rc) { printf("KSPLIT launch failed (n=%d form=%d): %s\n", n, form, usp_strerror(rc)); return 1; }
      HIP_OK(hipDeviceSynchronize());
      if (!check) continue;
      auto ob = dev_download(dout, nq); auto ab = dev_download(dacc, nq); auto lh = dev_download(dlse, nl);
      std::vector<float> of(nq);
      for (size_t i = 0; i < nq; ++i) {
        const int s = (int)((i / ((size_t)Hq * D)) % Sq);
        of[i] = s < a.final_end ? dec(ob[i], dt) : ab[i];
      }
      Err eo = compare(of.data(), ro.data(), nq, t.out_atol, t.out_rtol);
      Err el = compare(lh.data(), rl.data(), nl, t.lse_atol, 1e-4);
      const int bad = (eo.bad || el.bad || eo.nan);
      fail |= bad;
      printf("CHECK ksplit n=%d form=%d B%d Sq%d Sk%d Hq%d Hkv%d D%d %s %s : out max|err| %.3e bad %zu nan %zu | lse max|err| %.3e bad %zu  %s\n",
             n, form, B, Sq, Sk, Hq, Hkv, D, causal ? "causal" : "full", dt ? "fp16" : "bf16", eo.max_abs, eo.bad, eo.nan,
             el.max_abs, el.bad, bad ? "FAIL" : "ok");
    }
    if (iters > 0) {
      a.merge_in = 0; a.final_begin = 0; a.final_end = Sq; a.Sk = Sk; a.k.ptr = dk; a.v.ptr = dv;
      warm_up([&] { usp_flash_fwd(&a, nullptr); });
      HIP_OK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; ++i) usp_flash_fwd(&a, nullptr);
      HIP_OK(hipEventRecord(e1, 0)); HIP_OK(hipEventSynchronize(e1));
      float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
      printf("TIME  ksplit n=%d B%d Sq%d Sk%d Hq%d Hkv%d D%d %s  %.4f ms  %.0f TFLOP/s (incl. the merge launch)\n", n, B, Sq, Sk, Hq,
             Hkv, D, causal ? "causal" : "full", ms, attn_flops(B, Sq, Sk, Hq, D, causal) / (ms * 1e-3) / 1e12);
    }
  }
  hipFree(dq); hipFree(dk); hipFree(dv); hipFree(dout); hipFree(dacc); hipFree(dlse); hipFree(ws);
  return fail;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: kbench probe|fwd|fwdmerge|pieces|split|ksplit|bwd|suite|overlap [steps MiB workgroups] ...\n"); return 64; }
  std::string cmd = argv[1];
  auto I = [&](int i) { return atoi(argv[i]); };
  if (cmd == "probe") return run_probe();
  if (cmd == "suite") return suite(argc > 2 && std::string(argv[2]) == "bwd");
  if (cmd == "fwd" && argc >= 12) return run_fwd(I(2), I(3), I(4), I(5), I(6), I(7), I(8), I(9), I(10), I(11));
  if (cmd == "bwd" && argc >= 12) return run_bwd(I(2), I(3), I(4), I(5), I(6), I(7), I(8), I(9), I(10), I(11));
  if (cmd == "overlap") return run_overlap(argc > 2 ? I(2) : 4, argc > 3 ? I(3) : 16, argc > 4 ? I(4) : 8);
  if (cmd == "ksplit" && argc >= 12) return run_ksplit(I(2), I(3), I(4), I(5), I(6), I(7), I(8), I(9), I(10), I(11));
  if (cmd == "split") return run_split(argc > 2 ? I(2) : 16384, argc > 3 ? I(3) : 2, argc > 4 ? I(4) : 10);
  if (cmd == "pieces") return run_pieces(argc > 2 ? I(2) : 4096, argc > 3 ? I(3) : 4096, argc > 4 ? I(4) : 16, argc > 5 ? I(5) : 128, argc > 6 ? I(6) : 20);
  if (cmd == "fwdmerge" && argc >= 9) return run_fwdmerge(I(2), I(3), I(4), I(5), I(6), I(7), I(8));
  fprintf(stderr, "bad arguments\n");
  return 64;
}
