// Blockwise flash-attention backward for gfx950, dQ launch, ONE WAVE PER SIMD.  C ABI: usp_flash_bwd (include/usp_hip.h);
// replaces -- together with the dK/dV launch of usp_flash_bwd64.hip -- the reference's `bwd-only` block kernel
// (yunchang/kernels/attention.py:205-250).
//
// workgroup = 4 waves = 256 query rows of one (batch, head); a wave owns 64 rows (two 32-row query blocks) and its SIMD's
// whole register file:
//   a[0:127]   dQ^T accumulators, 2 query blocks x 4 dim tiles                       (asm MFMAs, "+a")
//   a[128:191] Q fragments, a[192:255] dO fragments of the wave's rows: B operands of the S^T / dP^T chains, never copied
//   v[...]     S^T -> P^T and dP^T -> dS^T of the streamed 64-key tile (2 x 64), packed dS^T (32), LDS fragments
// K / V tiles stream through LDS (LDS-DMA, two buffers each).  Everything is computed TRANSPOSED (keys down the rows of an
// accumulator tile, queries across the lanes), so that the row statistics lse / delta are ONE value per lane and query block
// and P, dS leave the accumulator layout as MFMA B operands without any cross-lane move (usp_flash_fwd64.hip).
// A tile is 96 MFMA slots per wave, in six blocks of 16, query block major, so that a block's elements can start while the
// other query block's MFMAs run:
//   B1  S^T[0] = K Q0^T    | B2  S^T[1] = K Q1^T    | B3  dP^T[0] = V dO0^T  | B4  dP^T[1] = V dO1^T
//   B5  dQ^T[0] += K^T dS^T[0]                      | B6  dQ^T[1] += K^T dS^T[1]
// Every LDS fragment is read once and serves two MFMAs (the second block of a pair takes it from the registers): 16 + 16
// row reads (ds_read_b128) and 32 transposed reads (ds_read_b64_tr_b16) per tile = 0.67 per MFMA.  The element streams
//   X[qb]: P = exp2(S c - lse)   (B2 / B3)        Y[qb]: dS = P (dP - delta), packed to 16 bits   (B4 / B5-B6)
// ride in the slots of the blocks named, the tile's 8 LDS-DMA pieces in B1.  2.25 VALU + 0.67 transcendental issues per
// MFMA: the stream stays under the five issue slots a lone wave hides per MFMA (profiles/r04_ubench_issue.txt), which the
// 64-MFMA tiles of the forward and of the dK/dV kernel do not.
// MFMAs are inline asm (usp_mfma64.hpp); tools/mfma_hazards.py checks the emitted stream.
#include <stdlib.h>

#include <type_traits>

#include "usp_bwd_params.hpp"
#include "usp_common.hpp"
#include "usp_hip.h"
#include "usp_mfma64.hpp"

namespace usp {

#ifndef USP_Q64_PF
#define USP_Q64_PF 3
#endif
constexpr int kQ64_PF = USP_Q64_PF;     // LDS fragments are read this many fragments ahead of the first MFMA that takes them
#ifndef USP_Q64_Y1
#define USP_Q64_Y1 24
#endif
constexpr int kQ64_Y1 = USP_Q64_Y1;    // gaps over which the last element stream (dS of query block 1) is spread, from B5 on
                               // (neither moves the kernel by more than 0.5 %: profiles/r04_run26*.log)

// dev build -DUSP_Q64_TIMING: where an item's time goes (s_memtime stamps, printed for a few waves)
#ifdef USP_Q64_TIMING
#define USP_TM(...) __VA_ARGS__
#else
#define USP_TM(...)
#endif

template <int DT, bool CAUSAL>
__global__ __launch_bounds__(256, 1) void flash_bwd_dq64_kernel(const BwdParams /* read through the kernarg segment */) {
  using E = Elem<DT>;
  using M = M64<DT>;
  constexpr int D = 128, kBM = 256;
  constexpr int ROWB = D * 2;
  constexpr int TILEB = kTile * ROWB;            // one K (or V) tile: 64 keys
  constexpr int VOFF = 2 * TILEB;                // LDS: Kbuf[0], Kbuf[1], Vbuf[0], Vbuf[1]
  constexpr int NKT = D / 16, NDJ = D / 32;
  constexpr int PF = kQ64_PF;

  // The dynamic LDS block is the kernel's only LDS object and starts at LDS address 0: addresses are formed from that
  // integer (hipcc does not fold the symbol's value and spends a v_add of 0 per address on it).
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if ((uint32_t)(uintptr_t)(USP_LDS char*)smem_raw != 0u) __builtin_trap();
  USP_LDS char* smem = (USP_LDS char*)(uintptr_t)0;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31;
  const int hi = lane >> 5;
  // the argument block stays in the kernarg segment (usp_flash_fwd64.hip: held in SGPRs it fills the scalar file)
  typedef const __attribute__((address_space(4))) BwdParams* KArgs;
  KArgs p = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));

  // ---- lane-constant addresses (tile layout, LDS-DMA pieces, row and transposed reads: as in usp_flash_bwd64.hip) ------
  // a tile is 4 groups of 16 rows, a group 4 pieces of 4 rows; wave w stages group w of the K tile and of the V tile
  const int dma_row = lane >> 4;
  const int dma_c8 = ((lane & 15) ^ ((lane >> 4) << 2)) * 16;
  const int k_voff = dma_row * (int)p->k_ss * 2 + dma_c8, v_voff = dma_row * (int)p->v_ss * 2 + dma_c8;
  const int rd_base = l31 * ROWB + ((hi ^ tile_swz<D>(l31)) * 16);            // ^ (32 kt), + 32 kb rows
  int tr_addr[NDJ][2];
  {
    const int i = lane & 15, grp = (lane >> 4) & 1;
#pragma unroll
    for (int dj = 0; dj < NDJ; ++dj)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int rr = 8 * e + 4 * hi + (i >> 2);
        const int slot = 4 * dj + 2 * grp + ((i & 3) >> 1);
        tr_addr[dj][e] = rr * ROWB + ((slot ^ tile_swz<D>(rr)) * 16) + (i & 1) * 8;
      }
  }
  const float c = p->scale_log2;

  const ItemWalk walk(p->n_items);               // persistent workgroups (usp_common.hpp)
  for (int pass = 0;; ++pass) {
  int w = walk.at(pass);
  if (w < 0) break;
  asm volatile("" : "+s"(p));
  USP_TM(const uint64_t tm_item = __builtin_amdgcn_s_memtime();)
  w = walk.dealt(w, p->nblk);
  if (p->ksplit <= 1) w = walk.grouped(w, p->nblk, p->walk_g);
  const int qt_r = w % p->nblk;
  int rest = w / p->nblk;
  const int qt = CAUSAL ? (p->nblk - 1 - qt_r) : qt_r;    // heavy (late) tiles first
  int cut = 0;                                            // key cut of a few-item launch (ABI v5 dq_splits): this item's run of
  if (p->ksplit > 1) { cut = rest % p->ksplit; rest /= p->ksplit; }     // the key tiles, partial to the workspace
  const int h = rest % p->Hq, b = rest / p->Hq;
  const int hkv = h / p->G;
  const int q0 = qt * kBM;
  const int qw = q0 + wave * 64;                          // first row of this wave
  const int off = p->causal_off;

  // ---- key range ---------------------------------------------------------------------------------------------------------
  int blk_kv_end = p->Sk, wave_kv_end = p->Sk;
  if (CAUSAL) {
    const int blk_last = (q0 + kBM < p->Sq ? q0 + kBM : p->Sq) - 1;
    const int wav_last = (qw + 64 < p->Sq ? qw + 64 : p->Sq) - 1;
    blk_kv_end = blk_last + off + 1 < p->Sk ? blk_last + off + 1 : p->Sk;
    wave_kv_end = wav_last + off + 1 < p->Sk ? wav_last + off + 1 : p->Sk;
  }
  if (qw >= p->Sq) wave_kv_end = 0;
  const int nt = blk_kv_end > 0 ? (blk_kv_end + kTile - 1) / kTile : 0;    // tiles the workgroup streams
  int n_full = p->Sk / kTile;                                               // leading tiles that need no mask for this wave
  if (CAUSAL) {
    const int lim = qw + off + 1;                          // keys < lim are visible to EVERY row of the wave
    const int nf = lim > 0 ? lim / kTile : 0;
    n_full = nf < n_full ? nf : n_full;
  }
  const int n_w = wave_kv_end > 0 ? (wave_kv_end + kTile - 1) / kTile : 0;  // tiles this wave works on
  if (n_full > n_w) n_full = n_w;
  // key cut: the workgroup streams tiles [tb, te) of its [0, nt) -- equal runs, as the 8-wave kernel cuts them
  int tb = 0, te = nt;
  if (p->ksplit > 1) {
    const int per = (nt + p->ksplit - 1) / p->ksplit;
    tb = cut * per < nt ? cut * per : nt;
    te = tb + per < nt ? tb + per : nt;
  }
  const int e_full = n_full < tb ? tb : (n_full > te ? te : n_full);        // [tb, e_full) plain, [e_full, e_own) masked,
  const int e_own = n_w < tb ? tb : (n_w > te ? te : n_w);                  // [e_own, te) other waves' tiles

  // ---- resident B operands: Q and dO fragments of the wave's 64 rows, loaded straight into the accumulator file (asm
  // loads + wait: usp_flash_bwd64.hip explains why hipcc must not see them) ------------------------------------------------
  u32x4 qf[2][NKT], df[2][NKT];
  float nl[2], dl[2];                            // -lse * log2(e) (-inf: the row sees no key -> P = 0) and delta per lane
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int row = qw + 32 * qb + l31;
    const int row_c = row < p->Sq ? row : p->Sq - 1;
    [[maybe_unused]] const char* qp = p->q + 2 * (b * p->q_sb + (int64_t)row_c * p->q_ss + h * p->q_sh) + 16 * hi;
    [[maybe_unused]] const char* dp = p->dout + 2 * (b * p->do_sb + (int64_t)row_c * p->do_ss + h * p->do_sh) + 16 * hi;
#if defined(__HIP_DEVICE_COMPILE__)
    // (16 loads, ONE wait: the four groups of an item cost two memory round trips, not four)
    asm volatile("global_load_dwordx4 %0, %16, off\n\tglobal_load_dwordx4 %1, %16, off offset:32\n\t"
                 "global_load_dwordx4 %2, %16, off offset:64\n\tglobal_load_dwordx4 %3, %16, off offset:96\n\t"
                 "global_load_dwordx4 %4, %16, off offset:128\n\tglobal_load_dwordx4 %5, %16, off offset:160\n\t"
                 "global_load_dwordx4 %6, %16, off offset:192\n\tglobal_load_dwordx4 %7, %16, off offset:224\n\t"
                 "global_load_dwordx4 %8, %17, off\n\tglobal_load_dwordx4 %9, %17, off offset:32\n\t"
                 "global_load_dwordx4 %10, %17, off offset:64\n\tglobal_load_dwordx4 %11, %17, off offset:96\n\t"
                 "global_load_dwordx4 %12, %17, off offset:128\n\tglobal_load_dwordx4 %13, %17, off offset:160\n\t"
                 "global_load_dwordx4 %14, %17, off offset:192\n\tglobal_load_dwordx4 %15, %17, off offset:224\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&a"(qf[qb][0]), "=&a"(qf[qb][1]), "=&a"(qf[qb][2]), "=&a"(qf[qb][3]), "=&a"(qf[qb][4]), "=&a"(qf[qb][5]),
                   "=&a"(qf[qb][6]), "=&a"(qf[qb][7]), "=&a"(df[qb][0]), "=&a"(df[qb][1]), "=&a"(df[qb][2]), "=&a"(df[qb][3]),
                   "=&a"(df[qb][4]), "=&a"(df[qb][5]), "=&a"(df[qb][6]), "=&a"(df[qb][7])
                 : "v"(qp), "v"(dp) : "memory");
#endif
    const float lse = p->lse[b * p->lse_sb + h * p->lse_sh + row_c];
    const float dlt = p->delta[b * p->dl_sb + h * p->dl_sh + row_c];
    const bool live = row < p->Sq && lse != USP_NEG_INF;
    nl[qb] = live ? -lse * kLog2e : USP_NEG_INF;
    dl[qb] = live ? dlt : 0.f;
  }

  // ---- LDS-DMA staging of the K / V tiles: running cursors at this wave's group (16 rows) of the tile ------------------
  const int k_rowb = (int)p->k_ss * 2, v_rowb = (int)p->v_ss * 2;
  const int64_t k_tb = (int64_t)kTile * k_rowb, v_tb = (int64_t)kTile * v_rowb;         // bytes per tile step
  int k_step = 4 * k_rowb - 1024, v_step = 4 * v_rowb - 1024;
  int lds_w = wave * 4096;
  const char* k_cur = p->k + 2 * (b * p->k_sb + hkv * p->k_sh) + (int64_t)wave * 16 * k_rowb + tb * k_tb;
  const char* v_cur = p->v + 2 * (b * p->v_sb + hkv * p->v_sh) + (int64_t)wave * 16 * v_rowb + tb * v_tb;
  int rows_kv = p->Sk - 16 * wave - tb * kTile;  // valid rows from the cursors on (<= 0: nothing left, lanes read 0)
  u32x4 k_rs, v_rs;
  int dma_buf = 0;
  auto dma_open = [&](int buf) {                 // scalar work only, no branch: it runs inside the MFMA stream
    k_rs = make_rsrc_rows(k_cur, rows_kv, 16, k_rowb, 2 * D);
    v_rs = make_rsrc_rows(v_cur, rows_kv, 16, v_rowb, 2 * D);
    dma_buf = buf;
    k_cur += k_tb;
    v_cur += v_tb;
    rows_kv -= kTile;
  };
  auto dma_piece = [&](int n) {                  // n < 4: K piece n, else V piece n - 4
    asm volatile("" : "+s"(lds_w), "+s"(k_step), "+s"(v_step));
    const int i = n & 3;
    if (n < 4) lds_dma16_asm(k_rs, lds_w + dma_buf * TILEB, k_voff ^ (16 * i), i * k_step, i);
    else lds_dma16_asm(v_rs, lds_w + VOFF + dma_buf * TILEB, v_voff ^ (16 * i), i * v_step, i);
  };

  f32x16 dq[2][NDJ];                             // dQ^T: [query block][dim tile]
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int dj = 0; dj < NDJ; ++dj) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[qb][dj][r] = 0.f;
      pin_agpr(dq[qb][dj]);
    }

  dma_open(tb & 1);                              // (iteration t reads buffer t & 1)
#pragma unroll
  for (int n = 0; n < 8; ++n) dma_piece(n);
  dma_drain();
  __syncthreads();

  // one iteration = one key tile: [stage the next tile] [the six blocks] [publish].  The tile body is STRAIGHT-LINE code
  // (usp_flash_bwd64.hip: anything conditional around the asm MFMAs makes hipcc copy accumulators); a tile that needs the
  // causal / ragged mask runs in a loop instance of its own (MASK) with the mask applied to S^T behind B1 and B2.
  auto iter = [&](auto mask_c, int t, bool work) __attribute__((always_inline)) {
    constexpr bool MASK = decltype(mask_c)::value;
    const int par = t & 1;
    if (work) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) { pin_agpr4(qf[qb][kt]); pin_agpr4(df[qb][kt]); }
      // buffer bases go INTO the swizzled offsets before the XOR (multiples of 256; the XOR touches bits 5-7)
      int kr = rd_base + par * TILEB, vr = rd_base + VOFF + par * TILEB;
      asm volatile("" : "+v"(kr), "+v"(vr));     // opaque per tile: hipcc otherwise hoists the XORed addresses and spills them
      USP_LDS const char* kt_base = smem + par * TILEB;
      f32x16 bs[2][2], bd[2][2];                 // S^T -> P^T, dP^T -> dS^T: [query block][key block]
      u32x4 pk[2][4];                            // packed dS^T: [query block][k-step of 16 keys]
      u32x4 ka[2 * NKT], va[2 * NKT], xa[4 * NDJ];
      auto rd_k = [&](int f) { ka[f] = *(USP_LDS const u32x4*)(smem + (f & 1) * 32 * ROWB + (kr ^ (32 * (f >> 1)))); };
      auto rd_v = [&](int f) { va[f] = *(USP_LDS const u32x4*)(smem + (f & 1) * 32 * ROWB + (vr ^ (32 * (f >> 1)))); };
      auto rd_x = [&](int f) {                   // fragment f = NDJ * ks + dj of K^T
        USP_LDS const char* xb = kt_base + (f / NDJ) * 16 * ROWB;
        const u32x2 a0 = lds_read_tr16(xb + tr_addr[f % NDJ][0]);
        const u32x2 a1 = lds_read_tr16(xb + tr_addr[f % NDJ][1]);
        xa[f] = u32x4{a0[0], a0[1], a1[0], a1[1]};
      };
      // element n (0 .. 31) of query block qb, in the order the dQ k-steps need them: n = 8 ks + r8 ->
      // [qb][ks >> 1][8 (ks & 1) + r8]
      auto X = [&](int qb, int n) {
        const int ks = n >> 3, kb = ks >> 1, r = 8 * (ks & 1) + (n & 7);
        bs[qb][kb][r] = fast_exp2(__builtin_fmaf(bs[qb][kb][r], c, nl[qb]));
      };
      auto Y = [&](int qb, int n) {
        const int ks = n >> 3, kb = ks >> 1, r = 8 * (ks & 1) + (n & 7);
        bd[qb][kb][r] = (bd[qb][kb][r] - dl[qb]) * bs[qb][kb][r];
        if (r & 1) pk[qb][ks][(n & 7) >> 1] = E::pack2(bd[qb][kb][r - 1], bd[qb][kb][r]);
      };
      auto mask = [&](int qb) {                  // key j is visible to query row i iff j <= min(i + off, Sk - 1)
        const int row = qw + 32 * qb + l31;
        int klim = p->Sk - 1;
        if (CAUSAL) klim = row + off < klim ? row + off : klim;
        const int kb0 = t * kTile + 4 * hi;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb0 + (r & 3) + 8 * (r >> 2);
          if (key > klim) bs[qb][0][r] = USP_NEG_INF;
          if (key + 32 > klim) bs[qb][1][r] = USP_NEG_INF;
        }
      };
      // ---------------- B1: S^T[0], the K row fragments arrive; the next tile's DMA ----------------
#pragma unroll
      for (int f = 0; f < PF; ++f) rd_k(f);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kt = i >> 1, kb = i & 1;
        if (kt == 0) M::template s_first<MASK>(bs[0][kb], ka[i], qf[0][0]);
        else M::template s_next<MASK>(bs[0][kb], ka[i], qf[0][kt]);
        if (i == 0) { __builtin_amdgcn_sched_barrier(0); dma_open(par ^ 1); }
        if (i + PF < 16) rd_k(i + PF);
        if (i >= 1 && i <= 8) dma_piece(i - 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MASK) { mfma_settle(bs[0]); mask(0); }
      // ---------------- B2: S^T[1] (fragments from the registers) | X[0]; the first V fragments ----------------
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kt = i >> 1, kb = i & 1;
        if (kt == 0) M::template s_first<MASK>(bs[1][kb], ka[i], qf[1][0]);
        else M::template s_next<MASK>(bs[1][kb], ka[i], qf[1][kt]);
        if (i == 0) __builtin_amdgcn_sched_barrier(0);     // the elements read B1's results: not in front of this MFMA
        X(0, 2 * i); X(0, 2 * i + 1);
        if (i >= 16 - PF) rd_v(i - (16 - PF));
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MASK) { mfma_settle(bs[1]); mask(1); }
      // ---------------- B3: dP^T[0], the V row fragments arrive | X[1] ----------------
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kt = i >> 1, kb = i & 1;
        if (kt == 0) M::template s_first<MASK>(bd[0][kb], va[i], df[0][0]);
        else M::template s_next<MASK>(bd[0][kb], va[i], df[0][kt]);
        if (i == 0) __builtin_amdgcn_sched_barrier(0);
        if (i + PF < 16) rd_v(i + PF);
        X(1, 2 * i); X(1, 2 * i + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---------------- B4: dP^T[1] | Y[0]; the first K^T fragments ----------------
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kt = i >> 1, kb = i & 1;
        if (kt == 0) M::template s_first<MASK>(bd[1][kb], va[i], df[1][0]);
        else M::template s_next<MASK>(bd[1][kb], va[i], df[1][kt]);
        if (i == 0) __builtin_amdgcn_sched_barrier(0);
        Y(0, 2 * i); Y(0, 2 * i + 1);
        if (i >= 16 - PF) rd_x(i - (16 - PF));
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---------------- B5: dQ^T[0], the K^T fragments arrive | B6: dQ^T[1] | Y[1] over the first Y1 gaps ----------------
      constexpr int NY1 = kQ64_Y1;
      static_assert(NY1 >= 16 && NY1 <= 28, "dS of k-step ks must be packed before gap 16 + 4 ks");
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        M::template o_acc<MASK>(dq[0][i % NDJ], xa[i], pk[0][i / NDJ]);
        if (i == 0) __builtin_amdgcn_sched_barrier(0);
        if (i + PF < 16) rd_x(i + PF);
#pragma unroll
        for (int n = i * 32 / NY1; n < (i + 1) * 32 / NY1; ++n) Y(1, n);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        M::template o_acc<MASK>(dq[1][i % NDJ], xa[i], pk[1][i / NDJ]);
        if (16 + i < NY1) {
#pragma unroll
          for (int n = (16 + i) * 32 / NY1; n < (16 + i + 1) * 32 / NY1; ++n) Y(1, n);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      dma_open(par ^ 1);
#pragma unroll
      for (int n = 0; n < 8; ++n) dma_piece(n);
    }
    dma_drain();            // this wave's pieces of the next tile have landed ...
    __syncthreads();        // ... and so have everybody else's; every wave is done with this tile's buffers
  };
  const std::integral_constant<bool, false> plain;
  const std::integral_constant<bool, true> masked;
  int t = tb;
  USP_TM(const uint64_t tm_loop = __builtin_amdgcn_s_memtime();)
  __builtin_amdgcn_s_waitcnt(0x0f70);            // (usp_flash_bwd64.hip: nothing may still count as pending at a loop header)
  for (; t < e_full; ++t) iter(plain, t, true);
  __builtin_amdgcn_s_waitcnt(0x0f70);
  for (; t < e_own; ++t) iter(masked, t, true);
  mfma_settle(dq);
  USP_TM(const uint64_t tm_own = __builtin_amdgcn_s_memtime(); const uint64_t tm_plain_n = n_full;)
  for (; t < te; ++t) iter(plain, t, false);     // tiles other waves of the workgroup still work on: keep the cadence
  USP_TM(const uint64_t tm_epi = __builtin_amdgcn_s_memtime();)

  // ---- epilogue: fp32 store / accumulate, or final 16-bit store (dQ^T: a lane holds 4-dim pieces of ONE query row) --------
  mfma_settle(dq);
  asm volatile("" : "+s"(p));
  const bool wide = (p->wide16 & 1) != 0;         // 16-bit final output, rows 16-byte aligned, nothing accumulated (never with a cut)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int row = qw + 32 * qb + l31;
    if (wide) {
      const int row_c = row < p->Sq ? row : 0;
      store_row16_wide<E, NDJ>(p->dq16 + 2 * (b * p->dq16_sb + (int64_t)row_c * p->dq16_ss + h * p->dq16_sh), dq[qb], p->scale, hi,
                               row < p->Sq);
    } else if (row < p->Sq) {
      float* o1 = p->dq + b * p->dq_sb + (int64_t)row * p->dq_ss + h * p->dq_sh;
      char* h1 = p->dq16 ? p->dq16 + 2 * (b * p->dq16_sb + (int64_t)row * p->dq16_ss + h * p->dq16_sh) : nullptr;
      int acc_f = p->accum_dq;
      if (p->ksplit > 1) {       // the partial of this cut, combined (deterministically) with the others and with an accumulated
        o1 = p->ws_dq + ((((int64_t)cut * p->B + b) * p->Sq + row) * p->Hq + h) * D;      // dq by reduce_cuts_kernel
        h1 = nullptr;
        acc_f = 0;
      }
      const float sc = p->scale;
#pragma unroll
      for (int dj = 0; dj < NDJ; ++dj)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int d0 = 32 * dj + 8 * g4 + 4 * hi;
          f32x4 v1 = {dq[qb][dj][4 * g4] * sc, dq[qb][dj][4 * g4 + 1] * sc, dq[qb][dj][4 * g4 + 2] * sc,
                      dq[qb][dj][4 * g4 + 3] * sc};
          if (acc_f) v1 += *(const f32x4*)(o1 + d0);
          if (h1) *(u32x2*)(h1 + 2 * d0) = u32x2{E::pack2(v1[0], v1[1]), E::pack2(v1[2], v1[3])};
          else *(f32x4*)(o1 + d0) = v1;
        }
    }
  }
  __syncthreads();          // the next item's prologue refills the tile buffers
USP_TM(
  if (pass < 3 && lane == 0 && (blockIdx.x % 61) == 0)
    printf("TQ wg %3d pass %d wave %d qt %2d tiles own %3d (plain %3d) wg %3d : prologue %6llu own tiles %8llu (%5llu / tile) idle %6llu epilogue %6llu\n",
           (int)blockIdx.x, pass, wave, qt, n_w, (int)tm_plain_n, nt, (unsigned long long)(tm_loop - tm_item),
           (unsigned long long)(tm_own - tm_loop), (unsigned long long)((tm_own - tm_loop) / (n_w > 0 ? n_w : 1)),
           (unsigned long long)(tm_epi - tm_own), (unsigned long long)(__builtin_amdgcn_s_memtime() - tm_epi));
)
  }  // next item
}

bool dq64_serves(const BwdParams& p_in) {
  // dense launches (bf16 / fp16) without a window or the dynamic item queue; the pieces' swizzle is XORed into the per-lane byte
  // offset (rows a multiple of 256 bytes apart), 64 rows of K / V span less than 2^31 bytes
  if (p_in.seq_q || p_in.seq_k || p_in.sched || p_in.win_on) return false;
  if ((p_in.k_ss * 2) % 256 != 0 || (p_in.v_ss * 2) % 256 != 0 || p_in.k_ss * 128 >= (1LL << 31) || p_in.v_ss * 128 >= (1LL << 31))
    return false;
  return true;
}

bool launch_dq64(const BwdParams& p_in, int dtype, bool causal, hipStream_t st, int* rc) {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  if (!dq64_serves(p_in)) return false;
  BwdParams p = p_in;
  p.nblk = (p.Sq + 255) / 256;
  p.n_items = p.B * p.Hq * p.nblk * p.ksplit;
  p.wide16 = (p.ksplit <= 1 && p.dq16 && !p.accum_dq && rows16_aligned(p.dq16, p.dq16_sb, p.dq16_ss, p.dq16_sh)) ? 1 : 0;
  const int grid = (!p.interleave && p.n_items > cus) ? cus : p.n_items;      // persistent: one workgroup per CU
  const size_t lds = 4 * kTile * 128 * 2;
  if (dtype == USP_BF16) {
    if (causal) hipLaunchKernelGGL((flash_bwd_dq64_kernel<0, true>), dim3(grid), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((flash_bwd_dq64_kernel<0, false>), dim3(grid), dim3(256), lds, st, p);
  } else {
    if (causal) hipLaunchKernelGGL((flash_bwd_dq64_kernel<1, true>), dim3(grid), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((flash_bwd_dq64_kernel<1, false>), dim3(grid), dim3(256), lds, st, p);
  }
  *rc = hipGetLastError() == hipSuccess ? USP_OK : USP_ELAUNCH;
  return true;
}

}  // namespace usp
