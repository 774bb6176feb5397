// Blockwise flash-attention backward for gfx950.  C ABI: usp_flash_bwd (include/usp_hip.h).
// Replaces the reference's `bwd-only` block kernel (yunchang/kernels/attention.py:205-250) plus the
// fp32 accumulation the ring schedules do on its results (zigzag_ring_flash_attn.py:147-170).
//
// Two launches, no atomics, deterministic:
//   dQ     (flash_bwd_kernel)      : workgroup = 8 waves x 32 query rows; streams K,V tiles (64 keys) through LDS.
//                                    lane owns a query row:  S^T = K Q^T, dP^T = V dO^T, dQ^T += K^T dS^T
//   dK, dV (flash_bwd_dkdv_kernel) : workgroup = 8 waves, 128 keys, two roles (below); the one-wave-per-SIMD form of it
//                                    lives in usp_flash_bwd64.hip and serves the dense D = 128 launches.
// Both are one engine: two LDS tiles X1,X2 (row-major, 16-byte-slot XOR swizzle chosen so that
// BOTH ds_read_b128 row reads and ds_read_b64_tr_b16 column reads are bank-conflict free), two
// register-resident fragment sets R1,R2, S = X1 R1^T, T = X2 R2^T, and tr-read "X^T" operands for
// the gradient MFMAs.  As in the forward, no cross-lane shuffle is needed for P / dS: the k-step
// order of the gradient MFMAs is defined as the order the S accumulator holds rows.
#include <stdlib.h>

#include <type_traits>

#include "usp_bwd_params.hpp"
#include "usp_common.hpp"
#include "usp_hip.h"

namespace usp {

template <int D, int DT, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void flash_bwd_kernel(
    const BwdParams p_in) {
  using E = Elem<DT>;
  constexpr int NT = 512;     // threads
  constexpr int OWN = (NT / 64) * 32;           // rows owned by the workgroup (256 q rows / 128 keys)
  constexpr int ROWB = D * 2;
  constexpr int TILEB = kTile * ROWB;           // one streamed matrix tile
  constexpr int BUFB = 2 * TILEB;
  constexpr int NKT = D / 16;
  constexpr int NDJ = D / 32;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  USP_LDS char* smem = (USP_LDS char*)smem_raw;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31;
  const int hi = lane >> 5;

  // ---- work items (persistent workgroups, usp_common.hpp ItemWalk) ------------------------------------
  const ItemWalk walk(p_in.n_items);
  ItemQueue queue{p_in.sched, p_in.seq_q, p_in.n_items / p_in.nblk, p_in.nblk,
                  p_in.Hq, OWN, CAUSAL ? 1 : 0};
  int qstate = 0;
  USP_LDS int* qslots = (USP_LDS int*)(smem + p_in.sched_lds);
  for (int pass = 0;; ++pass) {
  int w = p_in.sched ? item_queue_next(queue, qstate, qslots, pass) : walk.at(pass);
  if (w < 0) break;
  BwdParams p = p_in;
  if (!p_in.sched) w = walk.dealt(w, p.nblk);
  const int blk_r = w % p.nblk;
  int rest = w / p.nblk;
  int b, hkv, h0, blk, cut = 0;
  
    blk = CAUSAL ? (p.nblk - 1 - blk_r) : blk_r;          // late query blocks see most keys
    if (p.ksplit > 1) { cut = rest % p.ksplit; rest /= p.ksplit; }
    const int g = rest % p.G; rest /= p.G;
    hkv = rest % p.Hkv; b = rest / p.Hkv;
    h0 = hkv * p.G + g;
  
  int64_t ws_row0;
  if (!bind_sequence(p, b, &ws_row0)) continue;
  const int own0 = blk * OWN;                  // first owned row (query row / key)
  if (p.seq_q != nullptr && own0 >= (p.Sq)) continue;   // past the end of its sequence
  const int off = p.causal_off;
  const int ow = own0 + wave * 32;             // first row owned by this wave
  const int orow = ow + l31;                   // this lane's row
  const int own_len = p.Sq;
  const int orow_c = orow < own_len ? orow : own_len - 1;

  // ---- register-resident fragments R1, R2 (B operands: lane holds row[16t + 8hi .. +7]) -----------
  u32x4 r1[NKT], r2[NKT];
  {
    const char *p1, *p2;
    
      p1 = p.q + 2 * (b * p.q_sb + (int64_t)orow_c * p.q_ss + h0 * p.q_sh);
      p2 = p.dout + 2 * (b * p.do_sb + (int64_t)orow_c * p.do_ss + h0 * p.do_sh);
    
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      r1[t] = *(const u32x4*)(p1 + 32 * t + 16 * hi);
      r2[t] = *(const u32x4*)(p2 + 32 * t + 16 * hi);
    }
  }
  // lane-local row statistics
  float lse2_l = 0.f, delta_l = 0.f;
  
    const float l_ = p.lse[b * p.lse_sb + h0 * p.lse_sh + orow_c];
    lse2_l = (l_ == USP_NEG_INF) ? __builtin_inff() : l_ * kLog2e;
    delta_l = p.delta[b * p.dl_sb + h0 * p.dl_sh + orow_c];

  // ---- streamed range ---------------------------------------------------------------------------
  // key tiles [t_begin, t_end)
  const int str_len = p.Sk;
  int t_begin = 0, t_end = (str_len + kTile - 1) / kTile;     // tiles per head
  if (CAUSAL) {
    
      const int last = (own0 + OWN < p.Sq ? own0 + OWN : p.Sq) - 1;
      const int kv_end = last + off + 1 < p.Sk ? last + off + 1 : p.Sk;
      t_end = kv_end > 0 ? (kv_end + kTile - 1) / kTile : 0;
    
  }
  if (p.win_on) {                 // key tiles left of the window of the block's first row: not streamed
    const int first = own0 + p.win_lo;
    t_begin = first > 0 ? first / kTile : 0;
    if (t_begin > t_end) t_begin = t_end;
  }
  if (p.ksplit > 1) {             // this item's cut of the key tiles [t_begin, t_end): equal runs
    const int per = (t_end - t_begin + p.ksplit - 1) / p.ksplit;
    t_begin = t_begin + cut * per < t_end ? t_begin + cut * per : t_end;
    t_end = t_begin + per < t_end ? t_begin + per : t_end;
  }
  const int per_head = t_end - t_begin;
  const int n_iter = per_head;

  // ---- staging: LDS-DMA (buffer_load ... lds), no staging registers, no ds_write ---------------------
  // One wave-instruction fills 1 KiB of LDS linearly (wave-uniform base + lane*16), i.e. 1024/ROWB
  // whole tile rows.  The slot swizzle is therefore applied on the SOURCE: the lane that lands on
  // physical slot p of row r fetches logical slot p ^ swz(r) of that row (same 16-byte chunks of the
  // same row: coalescing is unaffected).  Rows past the end of the tensor read as 0 (descriptor
  // bounds); hipcc drains the DMA (vmcnt(0)) in front of the s_barrier that ends the iteration.
  constexpr int NW = NT / 64;                     // waves
  constexpr int CHUNKS = TILEB / 1024;            // 1 KiB pieces per matrix tile
  constexpr int CPW = (CHUNKS + NW - 1) / NW;     // pieces per wave per matrix
  constexpr int RPC = 1024 / ROWB;                // tile rows per piece
  int dma_voff1[CPW], dma_voff2[CPW];
  const int64_t ss1 = p.k_ss;
  const int64_t ss2 = p.v_ss;
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int cidx = wave + NW * i;
    const int r = cidx * RPC + lane / (D / 8);
    const int c8 = (lane % (D / 8)) ^ tile_swz<D>(r);
    dma_voff1[i] = r * (int)ss1 * 2 + c8 * 16;
    dma_voff2[i] = r * (int)ss2 * 2 + c8 * 16;
  }
  // Prefetch cursor: running 64-bit tile pointers / remaining-bytes counters, advanced by additions only.
  decltype(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, 0, 0, 0)) rs1, rs2;
  int dma_buf = 0;
  const int64_t tb1 = (int64_t)kTile * ss1 * 2, tb2 = (int64_t)kTile * ss2 * 2;   // bytes per tile step
  const char *pf_p1 = nullptr, *pf_p2 = nullptr;
  int64_t pf_rem1 = 0, pf_rem2 = 0;
  {                                              // base the cursor on tile t_begin of the K / V rows of (b, hkv)
    pf_p1 = p.k + 2 * (b * p.k_sb + hkv * p.k_sh) + t_begin * tb1;
    pf_p2 = p.v + 2 * (b * p.v_sb + hkv * p.v_sh) + t_begin * tb2;
    pf_rem1 = ((int64_t)(str_len - 1 - t_begin * kTile) * ss1 + D) * 2;
    pf_rem2 = ((int64_t)(str_len - 1 - t_begin * kTile) * ss2 + D) * 2;
  }
  // build the descriptors for the cursor's tile, then advance the cursor
  auto stage_setup = [&](int buf) {
    auto clampu = [](int64_t r) { return (int)(uint32_t)(r < 0 ? 0 : (r > 0xffffffffLL ? 0xffffffffLL : r)); };
    rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)pf_p1, 0, clampu(pf_rem1), 0x00020000);
    rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)pf_p2, 0, clampu(pf_rem2), 0x00020000);
    dma_buf = buf;
    pf_p1 += tb1; pf_p2 += tb2; pf_rem1 -= tb1; pf_rem2 -= tb2;
  };
  // piece pi in [0, 2*CPW): matrix pi & 1, chunk wave + NW * (pi >> 1)
  auto stage_piece = [&](int pi) {
    const int i = pi >> 1;
    const int cidx = wave + NW * i;
    if (CHUNKS % NW == 0 || cidx < CHUNKS) {
      USP_LDS char* d1 = smem + dma_buf * BUFB + cidx * 1024;
      if ((pi & 1) == 0) lds_dma16(rs1, d1, dma_voff1[i]);
      else lds_dma16(rs2, d1 + TILEB, dma_voff2[i]);
    }
  };
  auto stage_all = [&]() {
#pragma unroll
    for (int pi = 0; pi < 2 * CPW; ++pi) stage_piece(pi);
  };

  // ---- per-lane LDS read addresses ----------------------------------------------------------------
  // row read (A operand of S / T): tile row 32*n32 + l31, logical slot 2kt + hi
  const int rd_row = l31 * ROWB;
  const int rd_x = hi ^ tile_swz<D>(l31);
  // transpose read (A operand of the gradient MFMAs) for dim tile dj, element half e, k-step ks:
  // the 16-lane group reads the [4 rows][16 dims] block rows 16ks + 8e + 4hi + (0..3),
  // dims 32dj + 16*grp + (0..15); lane i supplies row i>>2, dims 4*(i&3)..+3.
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

  // ---- accumulators -----------------------------------------------------------------------------
  f32x16 acc1[NDJ];                      // dQ^T
#pragma unroll
  for (int dj = 0; dj < NDJ; ++dj)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[dj][r] = 0.f;  }
  
  const float c = p.scale_log2;

  if (n_iter > 0) { stage_setup(0); stage_all(); }
  dma_drain();            // this wave's DMA pieces of the staged tile have landed (usp_common.hpp)
  __syncthreads();

  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  constexpr int NST = 2 * NKT;                          // MFMAs of one S/T phase
  constexpr int NGR = 2 * NDJ;                          // MFMAs of one gradient phase

  int cur_tile = t_begin;                              // streamed tile of the current iteration
  for (int it = 0; it < n_iter; ++it) {
    const int buf = it & 1;
    const int tile = cur_tile;
    cur_tile = (cur_tile + 1 == t_end) ? t_begin : cur_tile + 1;
    const int s0 = tile * kTile;                       // first streamed row of this tile
    const bool prefetch = it + 1 < n_iter;
    if (prefetch) stage_setup(buf ^ 1);

    bool active = true, need_mask = false;
    
      // streamed = keys, owned = query rows
      int wave_kv_end = p.Sk;
      if (CAUSAL) {
        const int wl = (ow + 32 < p.Sq ? ow + 32 : p.Sq) - 1;
        wave_kv_end = wl + off + 1 < p.Sk ? wl + off + 1 : p.Sk;
      }
      active = ow < p.Sq && s0 < wave_kv_end && (!p.win_on || s0 + kTile - 1 >= ow + p.win_lo);
      need_mask = (s0 + kTile > p.Sk) || (CAUSAL && s0 + kTile - 1 > ow + off) || (p.win_on && s0 < ow + 31 + p.win_lo);

    if (active) {
      // Hand-pinned pipeline over the two 32-row halves h0, h1 of the tile (sched_barrier(0) fences;
      // hipcc otherwise emits MFMA clusters and VALU clusters):
      //   ST(h0) | ST(h1) || P,dS(h0) | GRAD(h0) || P,dS(h1) | GRAD(h1)
      // LDS operands are prefetched two MFMAs ahead; the first MFMA of a chain takes C = 0.
      USP_LDS const char* x1 = smem + buf * BUFB;
      USP_LDS const char* x2 = x1 + TILEB;
      f32x16 sS[2], sT[2];
      u32x4 pk_ds[2][2];

      auto apply_mask = [&](int h) {
        const int sr0 = s0 + 32 * h + 4 * hi;           // streamed row of register r: sr0 + 8(r>>2) + (r&3)
        
          int klim = p.Sk - 1;
          if (CAUSAL) klim = orow + off < klim ? orow + off : klim;
          const int klo = p.win_on ? orow + p.win_lo : -0x40000000;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = sr0 + (r & 3) + 8 * (r >> 2);
            if (key > klim || key < klo) sS[h][r] = USP_NEG_INF;
          }
        
      };
      // P and dS of element r of half h (+ pack when a pair completes)
      auto elem = [&](int h, int r) {
        float pr, ds;
        
          pr = fast_exp2(__builtin_fmaf(sS[h][r], c, -lse2_l));
          ds = pr * (sT[h][r] - delta_l);
        
        sS[h][r] = pr;
        sT[h][r] = ds;
        if (r & 1) {
          // pin_here: hipcc otherwise SINKS the whole element block of half 0 out of the S/T phase it is meant to
          // hide behind, into the block of its first use (the gradient phase) -- ~110 VALU in front of the
          // first gradient MFMA (seen in the .s; sched_barrier only pins the machine scheduler inside a block)
          uint32_t w = E::pack2(sT[h][r - 1], sT[h][r]);
          pin_here(w);
          pk_ds[h][r >> 3][(r & 7) >> 1] = w;
          
        }
      };
      // S/T phase of half h; `vh` >= 0: interleave the element work of half vh
      auto st_phase = [&](int h, int vh) {
        u32x4 f1[NKT], f2[NKT];
        auto rd = [&](int kt) {
          const int a = h * 32 * ROWB + rd_row + (((2 * kt) ^ rd_x) * 16);
          f1[kt] = *(USP_LDS const u32x4*)(x1 + a);
          f2[kt] = *(USP_LDS const u32x4*)(x2 + a);
        };
        rd(0);
        if (NKT > 1) rd(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sl = 0; sl < NST; ++sl) {
          const int kt = sl >> 1;
          if ((sl & 1) == 0) {
            if (kt + 2 < NKT) rd(kt + 2);
            sS[h] = E::mfma(f1[kt], r1[kt], kt == 0 ? zero16 : sS[h]);
          } else {
            sT[h] = E::mfma(f2[kt], r2[kt], kt == 0 ? zero16 : sT[h]);
          }
          if (vh >= 0) {
#pragma unroll
            for (int e = sl * 16 / NST; e < (sl + 1) * 16 / NST; ++e) elem(vh, e);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      // gradient phase of half h; `vh` >= 0: interleave the element work of half vh
      auto grad_phase = [&](int h, int vh) {
        u32x4 xa[NGR];
        auto rd = [&](int i) {                           // i -> (k2, dj): K^T fragments of the tile in x1
          const int k2 = i / NDJ, dj = i % NDJ;
          USP_LDS const char* xb = x1 + (2 * h + k2) * 16 * ROWB;
          const u32x2 a0 = lds_read_tr16(xb + tr_addr[dj][0]);
          const u32x2 a1 = lds_read_tr16(xb + tr_addr[dj][1]);
          xa[i] = u32x4{a0[0], a0[1], a1[0], a1[1]};
        };
        rd(0);
        if (NGR > 1) rd(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NGR; ++i) {
          if (i + 2 < NGR) rd(i + 2);
          const int k2 = i / NDJ, dj = i % NDJ;
          acc1[dj] = E::mfma(xa[i], pk_ds[h][k2], acc1[dj]);
          if (vh >= 0) {
#pragma unroll
            for (int e = i * 16 / NGR; e < (i + 1) * 16 / NGR; ++e) elem(vh, e);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };

      // the next tile's LDS-DMA pieces go out back to back in front of the first chain (spread over its MFMA slots they
      // measured +0.4 % here at two waves per SIMD, and +6 % in the dK/dV kernel)
      if (prefetch) stage_all();
      st_phase(0, -1);
      if (need_mask) apply_mask(0);
      st_phase(1, 0);
      if (need_mask) apply_mask(1);
      grad_phase(0, 1);
      grad_phase(1, -1);
    } else if (prefetch) {
      stage_all();
    }

    dma_drain();            // this wave's DMA pieces of the staged tile have landed (usp_common.hpp)
    __syncthreads();
  }

  // ---- epilogue: fp32 store / accumulate, or final 16-bit store ---------------------------------------
  if (orow < own_len) {
    float* o1;
    char* h1 = nullptr;                          // 16-bit final destination (row base), if any
    int acc_f1;
    if (p.ksplit > 1) {   // partial of this cut, combined (deterministically) by reduce_cuts_kernel
      o1 = p.ws_dq + ((((int64_t)cut * p.B + b) * p.Sq + orow) * p.Hq + h0) * D; acc_f1 = 0;
    } else {
      o1 = p.dq + b * p.dq_sb + (int64_t)orow * p.dq_ss + h0 * p.dq_sh; acc_f1 = p.accum_dq;
      if (p.dq16) h1 = p.dq16 + 2 * (b * p.dq16_sb + (int64_t)orow * p.dq16_ss + h0 * p.dq16_sh);
    }
#pragma unroll
    for (int dj = 0; dj < NDJ; ++dj)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int d0 = 32 * dj + 8 * g4 + 4 * hi;
        f32x4 v1 = {acc1[dj][4 * g4] * p.scale, acc1[dj][4 * g4 + 1] * p.scale,
                    acc1[dj][4 * g4 + 2] * p.scale, acc1[dj][4 * g4 + 3] * p.scale};
        if (acc_f1) v1 += *(const f32x4*)(o1 + d0);
        if (h1) *(u32x2*)(h1 + 2 * d0) = u32x2{E::pack2(v1[0], v1[1]), E::pack2(v1[2], v1[3])};
        else *(f32x4*)(o1 + d0) = v1;
        
      }
  }
  if (p_in.sched && p_in.interleave) break;   // one item per workgroup: leave room for other streams' kernels
  }  // next item
  if (p_in.sched && threadIdx.x == 0) item_queue_release(queue);
}

// ======================================================================================================
// dK/dV, role-specialised waves.
//
// A single-role wave needs K AND V fragments (64 regs) plus dK AND dV accumulators (128 regs) per wave: > 256
// registers, i.e. ONE wave per SIMD, and a lone wave can hide only ~5 instructions per MFMA (measured:
// 57 % of its cycles are active issue, MFMA pipe 32 % busy).  Here every 32-key slice is served by TWO
// waves that sit on the same SIMD (wave w and w + 4):
//   role A (waves 0-3): S = Q K^T -> P = exp2(S*c - lse)  -> dV^T += dO^T P      (K frags, dV acc)
//   role B (waves 4-7): dP = dO V^T, P from A, dS = P*(dP - delta) -> dK^T += Q^T dS (V frags, dK acc)
// A hands P (bf16, 4 KiB per 64x32 block, raw register image: lane-linear ds_write/ds_read_b128) to B
// through LDS; B runs ONE TILE BEHIND A, so the hand-off is ordered by the per-tile s_barrier that
// exists anyway (double-buffered P slots, triple-buffered Q/dO tiles).  No recompute: 32 MFMAs per
// wave and tile instead of 64, < 256 registers per wave, two waves per SIMD with complementary
// MFMA / transcendental mixes.
// ======================================================================================================
template <int D, int DT, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void flash_bwd_dkdv_kernel(const BwdParams p_in) {
  using E = Elem<DT>;
  constexpr int NW = 8, OWN = 128;
  constexpr int ROWB = D * 2;
  constexpr int TILEB = kTile * ROWB;
  constexpr int STATB = 2 * kTile * 4;
  constexpr int BUFB = 2 * TILEB + STATB;
  constexpr int NBUF = 3;
  constexpr int PSLOT = 4096;                    // P of one 64 x 32 block, 16-bit
  constexpr int POFF = NBUF * BUFB;              // P exchange: [4 slices][2 slots][PSLOT]
  constexpr int NKT = D / 16;
  constexpr int NDJ = D / 32;
  constexpr int CHUNKS = TILEB / 1024;
  constexpr int CPW = (CHUNKS + NW - 1) / NW;
  constexpr int RPC = 1024 / ROWB;
  constexpr int NGR = 2 * NDJ;                   // gradient MFMAs per half
  constexpr int PF = 2;                          // LDS operands are fetched this many MFMAs ahead

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  USP_LDS char* smem = (USP_LDS char*)smem_raw;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 2;                    // 0: A (S, P, dV)   1: B (dP, dS, dK)
  const int slice = wave & 3;
  const int l31 = lane & 31;
  const int hi = lane >> 5;

  const ItemWalk walk(p_in.n_items);             // persistent workgroups (usp_common.hpp)
  ItemQueue queue{p_in.sched, p_in.seq_k, p_in.n_items / p_in.nblk, p_in.nblk,
                  p_in.Hkv * p_in.ngrp, OWN, 0};
  int qstate = 0;
  USP_LDS int* qslots = (USP_LDS int*)(smem + p_in.sched_lds);
  for (int pass = 0;; ++pass) {
  int w = p_in.sched ? item_queue_next(queue, qstate, qslots, pass) : walk.at(pass);
  if (w < 0) break;
  BwdParams p = p_in;
  if (!p_in.sched) w = walk.dealt(w, p.nblk);
  const int blk = w % p.nblk;                    // early key blocks are seen by most rows: first
  int rest = w / p.nblk;
  int g = 0, cut = 0;
  if (p.qsplit > 1) { cut = rest % p.qsplit; rest /= p.qsplit; }
  if (p.ngrp > 1) { g = rest % p.ngrp; rest /= p.ngrp; }
  const int hkv = rest % p.Hkv, b = rest / p.Hkv;
  const int h0 = hkv * p.G + g * p.gsub;
  int64_t ws_row0;
  if (!bind_sequence(p, b, &ws_row0)) continue;
  const int own0 = blk * OWN;
  if (p.seq_q != nullptr && own0 >= p.Sk) continue;          // past the end of its sequence
  const int off = p.causal_off;
  const int ow = own0 + slice * 32;
  const int orow = ow + l31;
  const int orow_c = orow < p.Sk ? orow : p.Sk - 1;

  // K (role A) or V (role B) fragments of this wave's 32 keys
  u32x4 rf[NKT];
  {
    const char* pr = role == 0 ? p.k + 2 * (b * p.k_sb + (int64_t)orow_c * p.k_ss + hkv * p.k_sh)
                               : p.v + 2 * (b * p.v_sb + (int64_t)orow_c * p.v_ss + hkv * p.v_sh);
#pragma unroll
    for (int t = 0; t < NKT; ++t) rf[t] = *(const u32x4*)(pr + 32 * t + 16 * hi);
  }

  int t_begin = 0, t_end = (p.Sq + kTile - 1) / kTile;
  if (CAUSAL) {
    const int first_q = own0 - off > 0 ? own0 - off : 0;
    t_begin = first_q / kTile;
    if (t_begin > t_end) t_begin = t_end;
  }
  if (p.win_on) {                                // query rows beyond the window of the block's last key: not streamed
    const int last = own0 + OWN - 1 - p.win_lo;  // row i sees key j only if i <= j - win_lo
    const int te = last >= 0 ? last / kTile + 1 : 0;
    t_end = te < t_end ? te : t_end;
    if (t_begin > t_end) t_begin = t_end;
  }
  if (p.qsplit > 1) {                            // this item's cut of the query tiles [t_begin, t_end): equal runs
    const int per = (t_end - t_begin + p.qsplit - 1) / p.qsplit;
    t_begin = t_begin + cut * per < t_end ? t_begin + cut * per : t_end;
    t_end = t_begin + per < t_end ? t_begin + per : t_end;
  }
  const int per_head = t_end - t_begin;
  const int heads_here = p.gsub;
  const int n_iter = per_head * heads_here;

  // ---- LDS-DMA staging of the Q / dO tiles -----------------------------------------------------------------
  // ONE buffer descriptor pair per (item, query head), based on row 0 of that head; a tile is addressed by a scalar byte
  // offset (the instruction's soffset operand).  Per tile that is two s_add and the loads.  (Round 2 kept 64-bit tile
  // pointers / remaining-bytes counters and rebuilt both descriptors -- clamps included -- for every tile, and every
  // tile recomputed the 64-bit addresses of its row statistics: ~170 scalar instructions and 40-50 SGPR-spill reloads
  // (v_readlane) at the head of EVERY iteration of EVERY wave, 40 % of a wave's instruction stream -- the kernel was
  // bound by instruction issue, not by the MFMA pipe: with every element operation removed it still ran at 973 of
  // 1071 us, profiles/r03_bwd_ablations.txt.)  The host guarantees that a head's rows span less than 2^31 bytes.
  int dma_voff1[CPW], dma_voff2[CPW];
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int cidx = wave + NW * i;
    const int r = cidx * RPC + lane / (D / 8);
    const int c8 = (lane % (D / 8)) ^ tile_swz<D>(r);
    dma_voff1[i] = r * (int)p.q_ss * 2 + c8 * 16;
    dma_voff2[i] = r * (int)p.do_ss * 2 + c8 * 16;
  }
  float st_lse = 0.f, st_delta = 0.f;
  bool st_in = false;
  decltype(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, 0, 0, 0)) rs1, rs2;
  const int tb1 = kTile * (int)p.q_ss * 2, tb2 = kTile * (int)p.do_ss * 2;     // bytes per tile step
  int pf_tile = t_begin, pf_hh = 0, soff1 = 0, soff2 = 0;
  const float *lse_h = nullptr, *dl_h = nullptr;                             // row statistics of the cursor's head
  const bool stat_wave = wave == 4;              // a role-B wave fetches the tile's statistics: role A is the longer stream
  auto pf_head = [&]() {                         // (re)base the cursor on head h0 + pf_hh, tile t_begin
    auto clampu = [](int64_t r) { return (int)(uint32_t)(r < 0 ? 0 : (r > 0xffffffffLL ? 0xffffffffLL : r)); };
    const int h = h0 + pf_hh;
    rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.q + 2 * (b * p.q_sb + h * p.q_sh)), 0,
                                            clampu(((int64_t)(p.Sq - 1) * p.q_ss + D) * 2), 0x00020000);
    rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dout + 2 * (b * p.do_sb + h * p.do_sh)), 0,
                                            clampu(((int64_t)(p.Sq - 1) * p.do_ss + D) * 2), 0x00020000);
    lse_h = p.lse + b * p.lse_sb + h * p.lse_sh;
    dl_h = p.delta + b * p.dl_sb + h * p.dl_sh;
    pf_tile = t_begin;
    soff1 = t_begin * tb1;
    soff2 = t_begin * tb2;
  };
  pf_head();
  // issue the cursor's tile into LDS buffer `buf`, fetch its row statistics, advance the cursor
  auto stage_next = [&](int buf) {
    if (stat_wave) {                             // raw loads only: the values are consumed by stage_stats, a tile later
      const int r = pf_tile * kTile + lane;
      st_in = r < p.Sq;                          // rows past the end: P = 0 (their Q / dO rows read as zero)
      const int rc = st_in ? r : p.Sq - 1;
      st_lse = lse_h[rc];
      st_delta = dl_h[rc];
    }
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      const int cidx = wave + NW * i;
      if (CHUNKS % NW == 0 || cidx < CHUNKS) {
        USP_LDS char* d1 = smem + buf * BUFB + cidx * 1024;
        lds_dma16(rs1, d1, dma_voff1[i], soff1);
        lds_dma16(rs2, d1 + TILEB, dma_voff2[i], soff2);
      }
    }
    ++pf_tile;
    soff1 += tb1;
    soff2 += tb2;
    if (heads_here > 1 && pf_tile == t_end) { ++pf_hh; pf_head(); }
  };
  auto stage_stats = [&](int buf) {
    if (stat_wave) {
      const float l2 = (st_in && st_lse != USP_NEG_INF) ? st_lse * kLog2e : __builtin_inff();
      *(USP_LDS float*)(smem + buf * BUFB + 2 * TILEB + 4 * lane) = l2;
      *(USP_LDS float*)(smem + buf * BUFB + 2 * TILEB + 4 * kTile + 4 * lane) = st_in ? -st_delta : 0.f;   // NEGATED:
    }                                                                          // role B folds it into the dP chain
  };

  // ---- per-lane LDS addresses ------------------------------------------------------------------------
  const int rd_row = l31 * ROWB;
  const int rd_x = hi ^ tile_swz<D>(l31);
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
  USP_LDS char* pex = smem + POFF + slice * 2 * PSLOT + lane * 16;   // + slot*PSLOT + (2h+k2)*1024

  f32x16 acc[NDJ];                               // dV^T (role A) / dK^T (role B)
#pragma unroll
  for (int dj = 0; dj < NDJ; ++dj)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[dj][r] = 0.f;
  const float c = p.scale_log2;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  if (n_iter > 0) { stage_next(0); stage_stats(0); }
  dma_drain();            // this wave's DMA pieces of the staged tile have landed (usp_common.hpp)
  __syncthreads();

  // The streaming loop is instantiated once per role, with ROLE a compile-time constant, and the role
  // is chosen by ONE branch around the whole loop: both roles execute the same barrier sequence.  (With
  // run-time role tests inside the per-element code every MFMA slot was split into several basic
  // blocks: 5.9 SALU per MFMA and 51 % of wave cycles parked; with one loop holding both roles' tile
  // bodies their loop invariants added up and the kernel spilled.)
  auto stream = [&](auto role_c) {
    constexpr int ROLE = decltype(role_c)::value;
    int tile_a = t_begin;                          // streamed tile role A works on in iteration `it`
    int tile_b = t_begin;                          // ... and role B (the previous tile of A)
    int buf_a = 0, buf_b = NBUF - 1;               // LDS buffers of those tiles (it % 3, (it - 1) % 3)
    for (int it = 0; it <= n_iter; ++it) {
      const bool prefetch = it + 1 < n_iter;
      const int buf_n = buf_a + 1 == NBUF ? 0 : buf_a + 1;      // (it + 1) % NBUF
      if (prefetch) stage_next(buf_n);

      const int my_it = it - ROLE;
      const int buf_of_my = ROLE == 0 ? buf_a : buf_b;
      const int tile = ROLE == 0 ? tile_a : tile_b;
      tile_b = tile_a;
      tile_a = (tile_a + 1 == t_end) ? t_begin : tile_a + 1;
      const int s0 = tile * kTile;
      const bool valid = my_it >= 0 && my_it < n_iter;
      const bool active = valid && ow < p.Sk && (!CAUSAL || (s0 + kTile - 1 + off >= ow)) &&
                          (!p.win_on || s0 <= ow + 31 - p.win_lo);
      const bool need_mask = (CAUSAL && (s0 + off < ow + 31)) || (p.win_on && s0 + kTile - 1 > ow - p.win_lo);

      if (active) {
        {
          const int buf = buf_of_my;
          USP_LDS const char* x1 = smem + buf * BUFB;            // Q tile
          USP_LDS const char* x2 = x1 + TILEB;                   // dO tile
          USP_LDS const char* xs = ROLE == 0 ? x1 : x2;          // row-read operand of the S / dP chain
          USP_LDS const char* xg = ROLE == 0 ? x2 : x1;          // transpose-read operand of the gradient
          USP_LDS const char* stat = x1 + 2 * TILEB + (ROLE == 0 ? 0 : 4 * kTile);
          USP_LDS char* pslot = pex + (my_it & 1) * PSLOT;
          f32x16 sc[2];                                          // S (role A) / dP (role B) of the two halves
          u32x4 pk[2][2];                                        // packed P (A) / dS (B): B operand of the gradient
          u32x4 pin[2][2];                                       // role B: P received from A
          f32x4 st4;
          f32x4 stq[4];                                          // row statistics of one half, fetched ahead of use
          auto load_stat = [&](int h, int j) {
            stq[j] = *(USP_LDS const f32x4*)(stat + (32 * h + 4 * hi) * 4 + 32 * j);
          };
          auto load_stats = [&](int h) {
#pragma unroll
            for (int j = 0; j < 4; ++j) load_stat(h, j);
          };

          // element r of half h: role A: P = exp2(S*c - lse2); role B: dS = P * (dP - delta), where the dP chain
          // STARTS from -delta (the MFMA's C operand = the row statistics tuple: one VALU per score less in the role
          // that has the most of them)
          auto elem = [&](int h, int r) {
            if (ROLE == 0 && (r & 3) == 0) st4 = stq[r >> 2];
            float val;
            if (ROLE == 0) {
              val = fast_exp2(__builtin_fmaf(sc[h][r], c, -st4[r & 3]));
            } else {
              const uint32_t wd = pin[h][r >> 3][(r & 7) >> 1];
              const float pr = (r & 1) ? E::hi(wd) : E::lo(wd);
              val = pr * sc[h][r];
            }
            sc[h][r] = val;
            if (r & 1) pk[h][r >> 3][(r & 7) >> 1] = E::pack2(sc[h][r - 1], sc[h][r]);
            if (ROLE == 0 && (r & 7) == 7)                        // 8 elements done: hand one k-step of P to B
              *(USP_LDS u32x4*)(pslot + (2 * h + (r >> 3)) * 1024) = pk[h][r >> 3];
          };
          auto chain_phase = [&](int h, int vh) {
            u32x4 f[NKT];
            auto rd = [&](int kt) {
              f[kt] = *(USP_LDS const u32x4*)(xs + h * 32 * ROWB + rd_row + (((2 * kt) ^ rd_x) * 16));
            };
            f32x16 c0 = zero16;
            if (ROLE == 1) {                                     // -delta of this half's 16 rows: the chain's C operand
              load_stats(h);
#pragma unroll
              for (int r = 0; r < 16; ++r) c0[r] = stq[r >> 2][r & 3];
            }
#pragma unroll
            for (int kt = 0; kt < PF && kt < NKT; ++kt) rd(kt);
            if (ROLE == 1) {                                     // fetch A's P of this half early
              pin[h][0] = *(USP_LDS const u32x4*)(pslot + (2 * h) * 1024);
              pin[h][1] = *(USP_LDS const u32x4*)(pslot + (2 * h + 1) * 1024);
            }
            if (ROLE == 0 && vh >= 0) load_stats(vh);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
              if (kt + PF < NKT) rd(kt + PF);
              sc[h] = E::mfma(f[kt], rf[kt], kt == 0 ? c0 : sc[h]);
              if (vh >= 0) {
#pragma unroll
                for (int e = kt * 16 / NKT; e < (kt + 1) * 16 / NKT; ++e) elem(vh, e);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          };
          auto grad_phase = [&](int h, int vh) {
            u32x4 xa[NGR];
            auto rd = [&](int i) {
              const int k2 = i / NDJ, dj = i % NDJ;
              USP_LDS const char* xb = xg + (2 * h + k2) * 16 * ROWB;
              const u32x2 a0 = lds_read_tr16(xb + tr_addr[dj][0]);
              const u32x2 a1 = lds_read_tr16(xb + tr_addr[dj][1]);
              xa[i] = u32x4{a0[0], a0[1], a1[0], a1[1]};
            };
#pragma unroll
            for (int i = 0; i < PF && i < NGR; ++i) rd(i);
            if (ROLE == 0 && vh >= 0) load_stats(vh);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NGR; ++i) {
              if (i + PF < NGR) rd(i + PF);
              acc[i % NDJ] = E::mfma(xa[i], pk[h][i / NDJ], acc[i % NDJ]);
              if (vh >= 0) {
#pragma unroll
                for (int e = i * 16 / NGR; e < (i + 1) * 16 / NGR; ++e) elem(vh, e);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          };
          auto apply_mask = [&](int h) {                         // role A only: query row i sees key j iff j <= i + off
            if (CAUSAL) {
              const int d = orow - off - s0 - 4 * hi;            // one VGPR; thresholds are inline constants
#pragma unroll
              for (int r = 0; r < 16; ++r)
                if (d > 32 * h + (r & 3) + 8 * (r >> 2)) sc[h][r] = USP_NEG_INF;
            }
            if (p.win_on) {                                      // ... and only if j >= i + win_lo
              const int dl = orow - p.win_lo - s0 - 4 * hi;
#pragma unroll
              for (int r = 0; r < 16; ++r)
                if (dl < 32 * h + (r & 3) + 8 * (r >> 2)) sc[h][r] = USP_NEG_INF;
            }
          };

          chain_phase(0, -1);
          if (ROLE == 0 && need_mask) apply_mask(0);
          chain_phase(1, 0);
          if (ROLE == 0 && need_mask) apply_mask(1);
          grad_phase(0, 1);
          grad_phase(1, -1);
        }
      }

      if (prefetch) stage_stats(buf_n);
      buf_b = buf_a;
      buf_a = buf_n;
      dma_drain();            // this wave's DMA pieces of the staged tile have landed (usp_common.hpp)
      __syncthreads();
    }

  };
  if (role == 0) stream(std::integral_constant<int, 0>{});
  else stream(std::integral_constant<int, 1>{});

  // ---- epilogue ------------------------------------------------------------------------------------------
  if (orow < p.Sk) {
    float* o32;
    char* o16 = nullptr;
    int accf;
    const float mul = role == 0 ? 1.f : p.scale;
    if (p.split) {
      const int64_t wo = (((int64_t)(g * p.qsplit + cut) * p.ws_rows + ws_row0 + orow) * p.Hkv + hkv) * D;
      o32 = (role == 0 ? p.ws_dv : p.ws_dk) + wo; accf = 0;
    } else if (role == 0) {
      o32 = p.dv + b * p.dv_sb + (int64_t)orow * p.dv_ss + hkv * p.dv_sh; accf = p.accum_dv;
      if (p.dv16) o16 = p.dv16 + 2 * (b * p.dv16_sb + (int64_t)orow * p.dv16_ss + hkv * p.dv16_sh);
    } else {
      o32 = p.dk + b * p.dk_sb + (int64_t)orow * p.dk_ss + hkv * p.dk_sh; accf = p.accum_dk;
      if (p.dk16) o16 = p.dk16 + 2 * (b * p.dk16_sb + (int64_t)orow * p.dk16_ss + hkv * p.dk16_sh);
    }
#pragma unroll
    for (int dj = 0; dj < NDJ; ++dj)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int d0 = 32 * dj + 8 * g4 + 4 * hi;
        f32x4 v = {acc[dj][4 * g4] * mul, acc[dj][4 * g4 + 1] * mul, acc[dj][4 * g4 + 2] * mul,
                   acc[dj][4 * g4 + 3] * mul};
        if (accf) v += *(const f32x4*)(o32 + d0);
        if (o16) *(u32x2*)(o16 + 2 * d0) = u32x2{E::pack2(v[0], v[1]), E::pack2(v[2], v[3])};
        else *(f32x4*)(o32 + d0) = v;
      }
  }
  if (p_in.sched && p_in.interleave) break;   // one item per workgroup: leave room for other streams' kernels
  }  // next item
  if (p_in.sched && threadIdx.x == 0) item_queue_release(queue);
}

// dst[b,s,h,:] (+)= sum_g ws[g][row][h][:]   -- combines the per-query-head dK / dV partials.
// Dense: row = b*S + s.  Packed: (b, s) runs over B x max rows; sequence b owns rows first_b + s, s < rows_b.
template <int D, int DT>
__global__ __launch_bounds__(256) void reduce_heads_kernel(const BwdParams p) {
  using E = Elem<DT>;
  constexpr int C4 = D / 4;
  const int S = p.Sk, H = p.Hkv;
  const int64_t total = (int64_t)p.B * S * H * C4;
  const int64_t gstride = p.ws_rows * H * D;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const int64_t r = i / C4;
    const int h = (int)(r % H);
    const int64_t bs = r / H;
    const int sidx = (int)(bs % S);
    const int b = (int)(bs / S);
    int64_t row = sidx, wrow = bs, bb = b;       // row inside dk/dv (with b), row inside a workspace slab
    if (p.seq_k != nullptr) {
      if (sidx >= p.seq_k[2 * b + 1] || p.seq_q[2 * b + 1] <= 0) continue;
      row = wrow = p.seq_k[2 * b] + sidx;
      bb = 0;
    }
    const int64_t e = 4 * c4;
    float* pk = p.dk ? p.dk + bb * p.dk_sb + row * p.dk_ss + h * p.dk_sh + e : nullptr;
    float* pv = p.dv ? p.dv + bb * p.dv_sb + row * p.dv_ss + h * p.dv_sh + e : nullptr;
    f32x4 ak = p.accum_dk ? *(const f32x4*)pk : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 av = p.accum_dv ? *(const f32x4*)pv : f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t o = (wrow * H + h) * D + e;
    for (int g = 0; g < p.nslab; ++g) {
      ak += *(const f32x4*)(p.ws_dk + g * gstride + o);
      av += *(const f32x4*)(p.ws_dv + g * gstride + o);
    }
    if (p.dk16) {
      char* hk = p.dk16 + 2 * (bb * p.dk16_sb + row * p.dk16_ss + h * p.dk16_sh + e);
      *(u32x2*)hk = u32x2{E::pack2(ak[0], ak[1]), E::pack2(ak[2], ak[3])};
    } else {
      *(f32x4*)pk = ak;
    }
    if (p.dv16) {
      char* hv = p.dv16 + 2 * (bb * p.dv16_sb + row * p.dv16_ss + h * p.dv16_sh + e);
      *(u32x2*)hv = u32x2{E::pack2(av[0], av[1]), E::pack2(av[2], av[3])};
    } else {
      *(f32x4*)pv = av;
    }
  }
}

// dq[b,s,h,:] (+)= sum_cut ws_dq[cut][b][s][h][:]   -- combines the dQ partials of a key-cut launch (dense only).
template <int D, int DT>
__global__ __launch_bounds__(256) void reduce_cuts_kernel(const BwdParams p) {
  using E = Elem<DT>;
  constexpr int C4 = D / 4;
  const int64_t total = (int64_t)p.B * p.Sq * p.Hq * C4;
  const int64_t cstride = (int64_t)p.B * p.Sq * p.Hq * D;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    int64_t r = i / C4;
    const int h = (int)(r % p.Hq); r /= p.Hq;
    const int sidx = (int)(r % p.Sq);
    const int b = (int)(r / p.Sq);
    const int64_t e = 4 * c4;
    float* pq = p.dq ? p.dq + b * p.dq_sb + (int64_t)sidx * p.dq_ss + h * p.dq_sh + e : nullptr;
    f32x4 a = p.accum_dq ? *(const f32x4*)pq : f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t o = (((int64_t)b * p.Sq + sidx) * p.Hq + h) * D + e;
    for (int c = 0; c < p.ksplit; ++c) a += *(const f32x4*)(p.ws_dq + c * cstride + o);
    if (p.dq16) {
      char* hq = p.dq16 + 2 * (b * p.dq16_sb + (int64_t)sidx * p.dq16_ss + h * p.dq16_sh + e);
      *(u32x2*)hq = u32x2{E::pack2(a[0], a[1]), E::pack2(a[2], a[3])};
    } else {
      *(f32x4*)pq = a;
    }
  }
}

template <int D, int DT>
static int launch_bwd(BwdParams p, bool causal, hipStream_t st, int force, int skip) {
  constexpr size_t lds0 = 2 * (2 * kTile * D * 2);
  // dK,dV
  // persistent launches: one workgroup per CU (both kernels fit once per CU), each walks n_items / grid items
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  static const bool persist = [] { const char* e = getenv("USP_BWD_PERSIST"); return !(e && e[0] == '0'); }();
  p.nblk = (p.Sk + 127) / 128;
  p.n_items = p.B * p.Hkv * p.nblk * p.ngrp * p.qsplit;
  const bool pers = (persist || p.sched) && !p.interleave;
  int grid = (pers && p.n_items > cus) ? cus : p.n_items;
  const size_t qx = p.sched ? 16 : 0;            // LDS for the item queue's two slots
  // dK/dV: the one-wave-per-SIMD kernel (4 waves x 64 keys, usp_flash_bwd64.hip) where it applies; USP_BWD_WAVES=8 forces
  // the 8-wave kernel below
  // (per call: `force` = USP_FORCE_ROW64 / USP_FORCE_WAVE32, include/usp_hip.h)
  static const int forced_env = [] { const char* e = getenv("USP_BWD_WAVES"); return e ? atoi(e) : 0; }();
  const int forced_waves = (force & USP_FORCE_WAVE32) ? 8 : ((force & USP_FORCE_ROW64) ? 0 : forced_env);
  if ((force & USP_FORCE_ROW64) && !(D == 128 && ((skip & USP_BWD_SKIP_DKDV) || dkdv64_serves(p, DT)) && ((skip & USP_BWD_SKIP_DQ) || dq64_serves(p))))
    return USP_EUNSUPPORTED;     // (only the launches that will run have to be served)
  bool dkdv_done = (skip & USP_BWD_SKIP_DKDV) != 0;
  if (!dkdv_done && D == 128 && forced_waves != 8) {
    int rc64 = USP_ELAUNCH;
    if (launch_dkdv64(p, DT, causal, st, &rc64)) {
      if (rc64 != USP_OK) return rc64;
      dkdv_done = true;
      launch_kinds_note(USP_KIND_DKDV_ROW64);
    }
  }
  if (!dkdv_done) {
    // the 8-wave dK/dV kernel addresses the Q / dO tiles of a head by a 32-bit byte offset from the head's first row
    if ((int64_t)p.Sq * p.q_ss * 2 >= (1LL << 31) || (int64_t)p.Sq * p.do_ss * 2 >= (1LL << 31)) return USP_EUNSUPPORTED;
  {
    constexpr size_t lds2 = 3 * (2 * kTile * D * 2 + 2 * kTile * 4) + 4 * 2 * 4096;
    p.sched_lds = (int)lds2;
    if (causal)
      hipLaunchKernelGGL((flash_bwd_dkdv_kernel<D, DT, true>), dim3(grid), dim3(512), lds2 + qx, st, p);
    else
      hipLaunchKernelGGL((flash_bwd_dkdv_kernel<D, DT, false>), dim3(grid), dim3(512), lds2 + qx, st, p);
    launch_kinds_note(USP_KIND_DKDV_WAVE8);
  }
  }
  if (hipGetLastError() != hipSuccess) return USP_ELAUNCH;
  if (p.split && !(skip & USP_BWD_SKIP_DKDV)) {
    const int64_t items = (int64_t)p.B * p.Sk * p.Hkv * (D / 4);
    int64_t rg = (items + 255) / 256;
    rg = rg > 2048 ? 2048 : rg;
    hipLaunchKernelGGL((reduce_heads_kernel<D, DT>), dim3((int)rg), dim3(256), 0, st, p);
    if (hipGetLastError() != hipSuccess) return USP_ELAUNCH;
    launch_kinds_note(USP_KIND_REDUCE_HEADS);
  }
  if (skip & USP_BWD_SKIP_DQ) return USP_OK;
  // dQ: the one-wave-per-SIMD kernel (4 waves x 64 query rows, usp_flash_bwd_dq64.hip) where it applies
  static const int forced_dq_env = [] { const char* e = getenv("USP_BWD_DQ_WAVES"); return e ? atoi(e) : 0; }();
  const int forced_dq = force ? 0 : forced_dq_env;
  if (D == 128 && forced_waves != 8 && forced_dq != 8) {
    int rc64 = USP_ELAUNCH;
    if (launch_dq64(p, DT, causal, st, &rc64)) {
      if (rc64 != USP_OK) return rc64;
      launch_kinds_note(USP_KIND_DQ_ROW64);
      if (p.ksplit > 1) {          // same stream: the cuts' partials are complete when this starts
        const int64_t items = (int64_t)p.B * p.Sq * p.Hq * (D / 4);
        int64_t rg = (items + 255) / 256;
        rg = rg > 2048 ? 2048 : rg;
        hipLaunchKernelGGL((reduce_cuts_kernel<D, DT>), dim3((int)rg), dim3(256), 0, st, p);
        launch_kinds_note(USP_KIND_REDUCE_CUTS);
      }
      return hipGetLastError() == hipSuccess ? USP_OK : USP_ELAUNCH;
    }
  }
  p.nblk = (p.Sq + 255) / 256;
  p.n_items = p.B * p.Hq * p.nblk * p.ksplit;
  grid = (pers && p.n_items > cus) ? cus : p.n_items;
  p.sched_lds = (int)lds0;
  if (causal)
    hipLaunchKernelGGL((flash_bwd_kernel<D, DT, true>), dim3(grid), dim3(512), lds0 + qx, st, p);
  else
    hipLaunchKernelGGL((flash_bwd_kernel<D, DT, false>), dim3(grid), dim3(512), lds0 + qx, st, p);
  if (hipGetLastError() != hipSuccess) return USP_ELAUNCH;
  launch_kinds_note(USP_KIND_DQ_WAVE8);
  if (p.ksplit > 1) {            // same stream: the partials are complete when this starts
    const int64_t items = (int64_t)p.B * p.Sq * p.Hq * (D / 4);
    int64_t rg = (items + 255) / 256;
    rg = rg > 2048 ? 2048 : rg;
    hipLaunchKernelGGL((reduce_cuts_kernel<D, DT>), dim3((int)rg), dim3(256), 0, st, p);
    launch_kinds_note(USP_KIND_REDUCE_CUTS);
  }
  return hipGetLastError() == hipSuccess ? USP_OK : USP_ELAUNCH;
}

static bool ok16(const usp_tensor& t, int esize) {
  const int m = 16 / esize;
  return t.ptr && (reinterpret_cast<uintptr_t>(t.ptr) & 15) == 0 && t.stride_b % m == 0 &&
         t.stride_s % m == 0 && t.stride_h % m == 0;
}

}  // namespace usp

static int64_t ws_rows_of(const usp_bwd_args* a) {
  return (a->seq_q || a->seq_k) ? a->total_k : (int64_t)a->B * a->Sk;
}

static int cuts_of(int32_t n, bool packed) { return (packed || n < 2) ? 1 : (n > 8 ? 8 : n); }

static int device_cus() {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  return cus;
}

// Query heads of a KV group that ONE dK/dV work item streams into its accumulators (ABI v7: usp_bwd_args.dkdv_heads; a divisor
// of G = Hq / Hkv).  More heads per item: K / V fragments, their pre-scale and the epilogue once per run of heads, fewer fp32
// partial slabs (none when the item takes the whole group and nothing is cut) and a shorter reduce; fewer heads: more items,
// which is what balances a causal launch of few (batch, KV head, key block) triangles.  0 = the library decides, from what
// was measured on MI355X (profiles/r06_gqa_loop.txt; dK/dV launch + reduce alone, G = 8):
//   B1 S65536 H32/4 causal   55.1 (1) 54.8 (2) 54.7 (4) 55.2 (8) ms      one KV head of 16384 keys, causal  0.915 (1) 0.896 (2) 1.329 (4) ms
//   B1 S32768 H32/4 causal   14.07    13.91    13.86    14.05            two KV heads, 16384 keys, causal   1.847 (1) 1.798 (2) 1.776 (4) 2.713 (8)
//   one KV head, 8192 rows x 16384 keys, full  0.907 (1) 0.883 (2) 0.871 (4) 1.218 (8)
//   HBM fetch per launch at B1 S65536 (second box: 52.34 / 52.03 / 52.19 ms):  11.4 GB (1)  13.6 GB (2)  33.5 GB (4)
// -> the largest divisor that leaves two items per CU (causal; one when every item is equally long), capped at 4 (eight heads
// per item lose to four even with plenty of items: the per-item saving halves again while every CU's static list shortens to a
// handful of long items) and, for CAUSAL launches, at 2: the 32 workgroups an XCD runs side by side hold 32 consecutive key
// blocks, whose first visible tiles lie up to 62 tiles apart; head after head inside an item that lead adds up, and from three
// heads on the window of Q / dO tiles they stream together (62 tiles x 32 KiB x heads) no longer fits the XCD's 4 MiB L2 -- the
// fetch triples for a time gain inside the noise.  Packed batches keep 1.  USP_BWD_GSUB=n (read once) overrides the automatic
// choice for A/B runs.
static int dkdv_heads_of(const usp_bwd_args* a) {
  const int G = a->Hq / a->Hkv;
  if (G <= 1) return 1;
  if (a->dkdv_heads > 0) return (G % a->dkdv_heads == 0) ? a->dkdv_heads : -1;
  if (a->seq_q || a->seq_k) return 1;
  static const int forced = [] { const char* e = getenv("USP_BWD_GSUB"); return e ? atoi(e) : 0; }();
  if (forced > 0) {
    int g = forced > G ? G : forced;
    while (G % g != 0) --g;
    return g;
  }
  const int64_t base = (int64_t)a->B * a->Hkv * ((a->Sk + 127) / 128) * cuts_of(a->dkdv_splits, false);
  const bool triangles = a->causal || ((a->flags & USP_ATTN_WINDOW) && a->window_right >= 0);
  const int64_t want = (triangles ? 2LL : 1LL) * device_cus();
  int best = 1;
  const int cap = triangles ? 2 : 4;
  for (int g = 2; g <= G && g <= cap; ++g)
    if (G % g == 0 && base * (G / g) >= want) best = g;
  return best;
}

// [dK partials | dV partials | dQ partials]: ((G / dkdv_heads) * dkdv_splits) slabs of ws_rows x Hkv x D each for dK and for dV
// (none when that is one slab: the whole group in one item and no cut), dq_splits slabs of B x Sq x Hq x D for dQ (none without a cut)
static int64_t dkdv_part_bytes(const usp_bwd_args* a) {
  const int gsub = dkdv_heads_of(a);
  if (gsub < 1) return 0;
  const int64_t slabs = (int64_t)((a->Hq / a->Hkv) / gsub) * cuts_of(a->dkdv_splits, a->seq_q || a->seq_k);
  return slabs > 1 ? 2 * slabs * ws_rows_of(a) * a->Hkv * a->D * 4 : 0;
}

extern "C" int64_t usp_flash_bwd_workspace_bytes(const usp_bwd_args* a) {
  if (!a || a->Hkv <= 0 || a->Hq < a->Hkv || a->Hq % a->Hkv != 0) return 0;
  const int nq = cuts_of(a->dq_splits, a->seq_q || a->seq_k);
  return dkdv_part_bytes(a) + (nq > 1 ? (int64_t)nq * a->B * a->Sq * a->Hq * a->D * 4 : 0);
}

extern "C" int usp_flash_bwd(const usp_bwd_args* a, void* stream) {
  using namespace usp;
  launch_kinds_reset();
  if (!a || !a->lse || !a->delta) return USP_EINVAL;
  const int force = a->flags & (USP_FORCE_ROW64 | USP_FORCE_WAVE32);
  if (force == (USP_FORCE_ROW64 | USP_FORCE_WAVE32)) return USP_EINVAL;
  const int skip = a->flags & (USP_BWD_SKIP_DQ | USP_BWD_SKIP_DKDV);
  if (skip == (USP_BWD_SKIP_DQ | USP_BWD_SKIP_DKDV)) return USP_EINVAL;
  if (a->dtype != USP_BF16 && a->dtype != USP_FP16) return USP_EINVAL;
  if (a->B <= 0 || a->Sq <= 0 || a->Sk <= 0 || a->Hq <= 0 || a->Hkv <= 0) return USP_EINVAL;
  if (!(a->softmax_scale > 0.f)) return USP_EINVAL;
  if (a->D != 32 && a->D != 64 && a->D != 128) return USP_EUNSUPPORTED;
  if (a->Hq % a->Hkv != 0) return USP_EUNSUPPORTED;
  if (!a->dout.ptr || !a->q.ptr || !a->k.ptr || !a->v.ptr) return USP_EINVAL;
  const bool packed = a->seq_q != nullptr || a->seq_k != nullptr;
  if (packed && !(a->seq_q && a->seq_k && a->total_k > 0)) return USP_EINVAL;
  // an fp32 tensor may be absent only if its 16-bit final output is given and nothing is accumulated
  auto need32 = [](const usp_tensor& t32, const usp_tensor& t16, int accum) { return !t16.ptr || accum; };
  const bool want_dq = !(a->flags & USP_BWD_SKIP_DQ), want_dkdv = !(a->flags & USP_BWD_SKIP_DKDV);   // (a skipped launch needs no outputs)
  if ((want_dq && need32(a->dq, a->dq16, a->accum_dq) && !a->dq.ptr) || (want_dkdv && need32(a->dk, a->dk16, a->accum_dk) && !a->dk.ptr) ||
      (want_dkdv && need32(a->dv, a->dv16, a->accum_dv) && !a->dv.ptr))
    return USP_EINVAL;
  auto ok32 = [](const usp_tensor& t) { return !t.ptr || ok16(t, 4); };
  auto okh = [](const usp_tensor& t) {
    return !t.ptr || ((reinterpret_cast<uintptr_t>(t.ptr) & 7) == 0 && t.stride_b % 4 == 0 &&
                      t.stride_s % 4 == 0 && t.stride_h % 4 == 0);
  };
  if (!ok16(a->dout, 2) || !ok16(a->q, 2) || !ok16(a->k, 2) || !ok16(a->v, 2) || !ok32(a->dq) ||
      !ok32(a->dk) || !ok32(a->dv) || !okh(a->dq16) || !okh(a->dk16) || !okh(a->dv16))
    return USP_EUNSUPPORTED;
  // sliding window (flash-attn's window_size), as in usp_flash_fwd: causal caps the right bound at 0, a right bound is
  // the causal limit with a shifted offset, a left bound is a second mask term + a shorter streamed range
  const bool has_win = (a->flags & USP_ATTN_WINDOW) != 0;
  const int wl = has_win ? a->window_left : -1;
  const int wr = a->causal ? 0 : (has_win ? a->window_right : -1);
  if ((a->seq_q || a->seq_k) && (wl >= 0 || wr > 0)) return USP_EUNSUPPORTED;       // dense launches only
  if (a->dq_splits < 0 || a->dq_splits > 8 || a->dkdv_splits < 0 || a->dkdv_splits > 8) return USP_EINVAL;
  if (a->dkdv_heads < 0 || dkdv_heads_of(a) < 1) return USP_EINVAL;       // (not a divisor of Hq / Hkv)
  BwdParams p;
  p.dout = (const char*)a->dout.ptr; p.q = (const char*)a->q.ptr;
  p.k = (const char*)a->k.ptr; p.v = (const char*)a->v.ptr;
  p.lse = a->lse; p.delta = a->delta;
  p.dq = (float*)a->dq.ptr; p.dk = (float*)a->dk.ptr; p.dv = (float*)a->dv.ptr;
  p.do_sb = a->dout.stride_b; p.do_ss = a->dout.stride_s; p.do_sh = a->dout.stride_h;
  p.q_sb = a->q.stride_b; p.q_ss = a->q.stride_s; p.q_sh = a->q.stride_h;
  p.k_sb = a->k.stride_b; p.k_ss = a->k.stride_s; p.k_sh = a->k.stride_h;
  p.v_sb = a->v.stride_b; p.v_ss = a->v.stride_s; p.v_sh = a->v.stride_h;
  p.lse_sb = a->lse_stride_b; p.lse_sh = a->lse_stride_h;
  p.dl_sb = a->delta_stride_b; p.dl_sh = a->delta_stride_h;
  p.dq_sb = a->dq.stride_b; p.dq_ss = a->dq.stride_s; p.dq_sh = a->dq.stride_h;
  p.dk_sb = a->dk.stride_b; p.dk_ss = a->dk.stride_s; p.dk_sh = a->dk.stride_h;
  p.dv_sb = a->dv.stride_b; p.dv_ss = a->dv.stride_s; p.dv_sh = a->dv.stride_h;
  p.B = a->B; p.Sq = a->Sq; p.Sk = a->Sk; p.Hq = a->Hq; p.Hkv = a->Hkv; p.G = a->Hq / a->Hkv;
  p.nblk = 0;
  p.causal_off = a->Sk - a->Sq + (wr > 0 ? wr : 0);
  p.win_on = wl >= 0 ? 1 : 0; p.win_lo = a->Sk - a->Sq - (wl >= 0 ? wl : 0);
  p.scale = a->softmax_scale;
  p.scale_log2 = a->softmax_scale * kLog2e;
  p.accum_dq = a->accum_dq ? 1 : 0; p.accum_dk = a->accum_dk ? 1 : 0; p.accum_dv = a->accum_dv ? 1 : 0;
  p.dq16 = (char*)a->dq16.ptr; p.dk16 = (char*)a->dk16.ptr; p.dv16 = (char*)a->dv16.ptr;
  p.dq16_sb = a->dq16.stride_b; p.dq16_ss = a->dq16.stride_s; p.dq16_sh = a->dq16.stride_h;
  p.dk16_sb = a->dk16.stride_b; p.dk16_ss = a->dk16.stride_s; p.dk16_sh = a->dk16.stride_h;
  p.dv16_sb = a->dv16.stride_b; p.dv16_ss = a->dv16.stride_s; p.dv16_sh = a->dv16.stride_h;
  // GQA head split: with a workspace, every query head of a KV group gets its own workgroups and the
  // per-head partials are summed afterwards; without one the group's heads are looped inside a workgroup.
  p.seq_q = a->seq_q; p.seq_k = a->seq_k;
  p.sched = packed ? a->sched : nullptr;
  p.sched_lds = 0;
  p.interleave = (a->flags & USP_LAUNCH_INTERLEAVE) ? 1 : 0;
  {   // USP_ITEM_GROUP=0 (read once): the head-major item walk of rounds 1-5 instead of a KV group's heads side by side
    static const bool group_heads = [] { const char* e = getenv("USP_ITEM_GROUP"); return !(e && e[0] == '0'); }();
    p.walk_g = group_heads ? p.G : 1;
  }
  p.wide16 = 0;                                   // (set by the 64-row launches for their own copy)
  p.ws_rows = ws_rows_of(a);
  if (packed) {
    p.do_sb = p.q_sb = p.k_sb = p.v_sb = p.lse_sb = p.dl_sb = 0;
    p.dq_sb = p.dk_sb = p.dv_sb = p.dq16_sb = p.dk16_sb = p.dv16_sb = 0;
  }
  // Workspace present and large enough: the dK/dV items of dkdv_heads_of() query heads each and / or the requested cuts;
  // otherwise neither (the whole KV group inside one workgroup, one item per key block) -- results are identical up to fp32
  // summation order either way.
  const int64_t need = usp_flash_bwd_workspace_bytes(a);
  p.split = 0; p.qsplit = 1; p.ksplit = 1; p.nslab = 1; p.ws_dk = nullptr; p.ws_dv = nullptr; p.ws_dq = nullptr;
  p.gsub = p.G; p.ngrp = 1;
  if (need > 0 && a->workspace && a->workspace_bytes >= need &&
      (reinterpret_cast<uintptr_t>(a->workspace) & 15) == 0) {
    const int64_t part = dkdv_part_bytes(a);
    if (part > 0) {
      p.split = 1;
      p.gsub = dkdv_heads_of(a);
      p.ngrp = p.G / p.gsub;
      p.qsplit = cuts_of(a->dkdv_splits, packed);
      p.nslab = p.ngrp * p.qsplit;
      p.ws_dk = (float*)a->workspace;
      p.ws_dv = p.ws_dk + part / 8;
    }
    p.ksplit = cuts_of(a->dq_splits, packed);
    if (p.ksplit > 1) p.ws_dq = (float*)((char*)a->workspace + part);
  }
  hipStream_t st = (hipStream_t)stream;
  const bool causal = wr >= 0;                    // (a->causal, or a right window bound)
  switch (a->D * 2 + a->dtype) {
    case 64: return launch_bwd<32, 0>(p, causal, st, force, skip);
    case 65: return launch_bwd<32, 1>(p, causal, st, force, skip);
    case 128: return launch_bwd<64, 0>(p, causal, st, force, skip);
    case 129: return launch_bwd<64, 1>(p, causal, st, force, skip);
    case 256: return launch_bwd<128, 0>(p, causal, st, force, skip);
    case 257: return launch_bwd<128, 1>(p, causal, st, force, skip);
  }
  return USP_EUNSUPPORTED;
}
