// Argument block and LDS-layout helpers shared by the backward kernels (usp_flash_bwd.hip: 8 waves, two per SIMD;
// usp_flash_bwd64.hip: 4 waves, one per SIMD, 64 owned rows per wave).
#pragma once
#include "usp_common.hpp"

namespace usp {

struct BwdParams {
  const char* dout; const char* q; const char* k; const char* v;
  const float* lse; const float* delta;
  float* dq; float* dk; float* dv;
  int64_t do_sb, do_ss, do_sh;
  int64_t q_sb, q_ss, q_sh;
  int64_t k_sb, k_ss, k_sh;
  int64_t v_sb, v_ss, v_sh;
  int64_t lse_sb, lse_sh, dl_sb, dl_sh;
  int64_t dq_sb, dq_ss, dq_sh;
  int64_t dk_sb, dk_ss, dk_sh;
  int64_t dv_sb, dv_ss, dv_sh;
  int B, Sq, Sk, Hq, Hkv, G, nblk;   // nblk = blocks along the owned sequence
  int n_items;                        // work items of this launch (persistent workgroups walk them)
  int causal_off;
  float scale, scale_log2;
  int accum_dq, accum_dk, accum_dv;
  char* dq16; char* dk16; char* dv16;   // optional 16-bit final outputs
  int64_t dq16_sb, dq16_ss, dq16_sh, dk16_sb, dk16_ss, dk16_sh, dv16_sb, dv16_ss, dv16_sh;
  float* ws_dk; float* ws_dv;        // head-split partials [G][ws_rows][Hkv][D] fp32 (G > 1)
  int64_t ws_rows;                    // key rows per head-group slab: B*Sk, packed mode: rows of k
  int split;                          // 1: the dK/dV items write fp32 partials to the workspace (nslab > 1), reduce_heads_kernel sums them
  int gsub, ngrp;                     // dK/dV launch: an item streams `gsub` query heads of its KV group (a divisor of G) into the same
                                      // accumulators; ngrp = G / gsub items per (KV head, key block).  gsub == G: the whole group inside the
                                      // workgroup (no partials unless cut); gsub == 1: one item per query head (round 1's "head split")
  int qsplit;                         // dK/dV launch: every (head, key block) item is cut into this many items over equal
                                      // runs of the query tiles it sees (>= 1; > 1 only with `split`)
  int ksplit;                         // dQ launch: every (head, query block) item is cut along the key tiles it sees
  int nslab;                          // dK/dV partial slabs the reduce sums: ngrp * qsplit
  float* ws_dq;                       // dQ partials [ksplit][B*Sq][Hq][D] fp32 (ksplit > 1)
  int win_on, win_lo;                 // sliding window, left bound (ABI v5): query row i sees key j only if
                                      // j >= i + win_lo (= Sk - Sq - window_left); the right bound is the causal limit
                                      // with a shifted offset (causal_off = Sk - Sq + window_right)
  const int* seq_q; const int* seq_k; // packed variable-length batch: B (first row, rows) pairs, or NULL
  int* sched;                         // packed mode: control block of the dynamic item queue, or NULL
  int sched_lds;                      // byte offset of the queue's two LDS slots
  int interleave;                     // USP_LAUNCH_INTERLEAVE: one workgroup per item (collectives can slip in)
  int walk_g;                         // 64-row dQ kernel: query heads of a KV group walked side by side (usp_item_deal.h); 1 = off
  int wide16;                         // 64-row kernels: bit 0 / 1 / 2 = the dq16 / dk16 / dv16 rows are 16-byte aligned and nothing
                                      // is accumulated into them: whole-row-piece stores (usp_mfma64.hpp: store_row16_wide)
};

// rows of a 16-bit output tensor start on 16-byte boundaries (base pointer and every stride)
inline bool rows16_aligned(const char* ptr, int64_t sb, int64_t ss, int64_t sh) {
  return ptr && (reinterpret_cast<uintptr_t>(ptr) & 15) == 0 && sb % 8 == 0 && ss % 8 == 0 && sh % 8 == 0;
}

// Packed variable-length batch: rebase the local copy of the parameters on the rows of sequence b (the
// host passes batch strides of 0 in this mode, so every `b * stride_b` vanishes).  Returns false if the
// sequence is empty on either side; *ws_row0 = first row of the sequence in the dK/dV workspace slabs.
USP_DEV bool bind_sequence(BwdParams& p, int b, int64_t* ws_row0) {
  if (p.seq_q == nullptr) { *ws_row0 = (int64_t)b * p.Sk; return true; }
  const int qf = p.seq_q[2 * b], ql = p.seq_q[2 * b + 1];
  const int kf = p.seq_k[2 * b], kl = p.seq_k[2 * b + 1];
  *ws_row0 = kf;
  if (ql <= 0 || kl <= 0) return false;
  p.dout += 2 * qf * p.do_ss; p.q += 2 * qf * p.q_ss;
  p.k += 2 * kf * p.k_ss; p.v += 2 * kf * p.v_ss;
  p.lse += qf; p.delta += qf;
  if (p.dq) p.dq += qf * p.dq_ss;
  if (p.dk) p.dk += kf * p.dk_ss;
  if (p.dv) p.dv += kf * p.dv_ss;
  if (p.dq16) p.dq16 += 2 * qf * p.dq16_ss;
  if (p.dk16) p.dk16 += 2 * kf * p.dk16_ss;
  if (p.dv16) p.dv16 += 2 * kf * p.dv16_ss;
  p.Sq = ql; p.Sk = kl; p.causal_off = kl - ql;
  return true;
}

constexpr int kTile = 64;           // streamed rows per LDS tile


// Swizzle of the 16-byte slot index inside a row-major [rows][D] 16-bit tile.
template <int D> USP_DEV int tile_swz(int row) {
  if (D == 128) return ((row & 3) << 2) | ((row >> 2) & 3);
  if (D == 64) return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
  return (row >> 2) & 3;   // D == 32
}

// The one-wave-per-SIMD dK/dV launch (usp_flash_bwd64.hip): dense launches of D = 128 without a window.  `p` is the
// complete argument block of the dK/dV launch (nblk / n_items are set inside).  Returns false when the launch is not
// one it serves (the caller then takes the 8-wave kernel).
bool launch_dkdv64(const BwdParams& p, int dtype, bool causal, hipStream_t st, int* rc);
bool dkdv64_serves(const BwdParams& p, int dtype);     // ... whether it would (no launch)
// The one-wave-per-SIMD dQ launch (usp_flash_bwd_dq64.hip): dense launches of D = 128 (bf16 / fp16) without a window; key cuts
// (ksplit > 1: partials to ws_dq, the caller launches reduce_cuts_kernel behind it) since round 5.
bool launch_dq64(const BwdParams& p, int dtype, bool causal, hipStream_t st, int* rc);
bool dq64_serves(const BwdParams& p);

}  // namespace usp
