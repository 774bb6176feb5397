// Argument blocks and LDS-layout constants shared by the forward kernels (usp_flash_fwd.hip: 8 / 4 waves x 32 query
// rows, two waves per SIMD; usp_flash_fwd64.hip: 4 waves x 64 query rows, one wave per SIMD).
#pragma once
#include "usp_common.hpp"

namespace usp {

struct FwdParams {
  const char* q; const char* k; const char* v;
  char* out; float* acc; float* lse;
  int64_t q_sb, q_ss, q_sh;
  int64_t k_sb, k_ss, k_sh;
  int64_t v_sb, v_ss, v_sh;
  int64_t o_sb, o_ss, o_sh;
  int64_t a_sb, a_ss, a_sh;
  int64_t lse_sb, lse_sh;
  int B, Sq, Sk, Hq, Hkv, G, nq, n_items;
  int causal_off;                 // Sk - Sq
  float scale, scale_log2;
  int merge_in, final_begin, final_end;
  int out_wide;                   // out rows are 16-byte aligned: 16-byte epilogue stores
  const int* seq_q; const int* seq_k;   // packed variable-length batch: B (first row, rows) pairs, or NULL
  int* sched;                           // packed mode: control block of the dynamic item queue, or NULL
  int interleave;                       // USP_LAUNCH_INTERLEAVE: one workgroup per item (collectives can slip in)
  int walk_g;                           // 64-row kernel: query heads of a KV group walked side by side (usp_item_deal.h: usp_group_item); 1 = off
};

// K split (dense mode): every (batch, head, query tile) is cut into `ksplit` items along K; partial results go to
// [ksplit][B,Sq,Hq,D] fp32 and [ksplit][B,Hq,Sq] fp32.  Kernel arguments of the split instantiation ONLY: the plain
// kernels keep the argument block (and with it the machine code) they were profiled with.
struct FwdSplit {
  int ksplit;
  float* ws_o; float* ws_lse;
  // Sliding window, left bound (ABI v5): query row i sees key j only if j >= i + win_lo (win_lo = Sk - Sq - window_left).
  // Lives in the split instantiation: tiles left of a query tile's window are skipped by the same rebasing the K split
  // uses, and every tile of a windowed launch takes the generic (masked) loop.  The RIGHT bound needs nothing new: it
  // is the causal limit with a shifted offset (host: causal_off = Sk - Sq + window_right, causal instantiation).
  int win_on, win_lo;
};
template <bool KS> struct FwdArgsT : FwdParams {};
template <> struct FwdArgsT<true> : FwdParams, FwdSplit {};

constexpr int kBN = 64;    // keys per KV tile

template <int D> struct KSwz {
  // 16-byte slots per K row and rows per 256-byte LDS bank row
  static constexpr int SPR = D / 8;
  static constexpr int RPB = 16 / SPR < 1 ? 1 : 16 / SPR;
  static USP_DEV int of(int row) { return (row / RPB) & (SPR - 1); }
};

// The 64-rows-per-wave forward (usp_flash_fwd64.hip): dense launches of D = 128, plain or K-split (no window, no packed
// batch).  Returns false when the launch is not one it serves (the caller then takes the 8-wave kernel).
bool launch_fwd64(const FwdArgsT<true>& p, int dtype, bool causal, hipStream_t st, int* rc);
// The merge launch behind a K-split forward (usp_flash_fwd.hip: split_merge_kernel), same stream.
int launch_split_merge(const FwdArgsT<true>& p, int dtype, int D, hipStream_t st);

}  // namespace usp
