// Shared device-side helpers for the gfx950 (CDNA4) kernels of libusp_hip.so.
// Wave = 64 lanes everywhere; MFMA shape used throughout is v_mfma_f32_32x32x16_{bf16,f16}:
//   A operand: lane l holds A[i = l&31][k = 8*(l>>5) + 0..7]      (8 x 16-bit = 4 VGPRs)
//   B operand: lane l holds B[k = 8*(l>>5) + 0..7][j = l&31]
//   C/D      : lane l, reg r holds D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31]
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#define USP_DEAL_FN __device__ __forceinline__
#include "usp_item_deal.h"

namespace usp {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

#define USP_LDS __attribute__((address_space(3)))
#define USP_DEV __device__ __forceinline__

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
#define USP_NEG_INF (-__builtin_inff())

// Element traits: DT = 0 bf16, 1 fp16.
template <int DT> struct Elem;
template <> struct Elem<0> {
  static USP_DEV f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  // Accumulate into an accumulator PINNED to the AGPR file ("+a").  hipcc otherwise gives long-lived
  // accumulators VGPR homes in >256-register kernels and copies them through AGPRs around every MFMA
  // (measured: 856 v_accvgpr moves per kernel).  Accumulate chains need no wait states; the leading
  // s_nop 1 covers "VALU write -> MFMA SrcA/B read", which hipcc does not pad inside asm.
  // NOTE: hipcc pads nothing inside asm.  Callers guarantee (by the pinned schedule) that a and b were
  // written at least one MFMA slot earlier ("VALU write -> MFMA SrcA/B read" needs 2 wait states).
  static USP_DEV void mfma_agpr(f32x16& acc, u32x4 a, u32x4 b) {
#if defined(__HIP_DEVICE_COMPILE__)   // "a" means eax to the host pass, which then drops the kernel stubs
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
#endif
  }
  static USP_DEV uint32_t pack2(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
  }
  static USP_DEV float lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
  static USP_DEV float hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
};
template <> struct Elem<1> {
  static USP_DEV f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static USP_DEV void mfma_agpr(f32x16& acc, u32x4 a, u32x4 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
#endif
  }
  static USP_DEV uint32_t pack2(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
  }
  static USP_DEV float lo(uint32_t w) {
    return (float)__builtin_bit_cast(f16x2, w)[0];
  }
  static USP_DEV float hi(uint32_t w) {
    return (float)__builtin_bit_cast(f16x2, w)[1];
  }
};

// Exchange a value with the other half-wave (lane l <-> lane l^32).
// v_permlane32_swap(vdst, src): lanes 32-63 of vdst swap with lanes 0-31 of src, so with both
// operands holding x the results are r[0] = {x.lo | x.lo}, r[1] = {x.hi | x.hi}.
// hipcc (ROCm 7.2) mis-folds the builtin's two results when they are consumed as floats
// (`r[0] + r[1]` was emitted as v_add_f32 r0, r0 -- reproduced in a 6-line kernel, also with an
// opaque copy), so the copy + swap are done in one asm statement.  The s_nop 1 covers the
// "VALU write -> v_permlane read" hazard (2 wait states) that hipcc does not pad inside asm.
USP_DEV void xhalf_pair(float x, float& lo, float& hi) {
  uint32_t a = __builtin_bit_cast(uint32_t, x), b;
  asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "=&v"(b));
  lo = __builtin_bit_cast(float, a);
  hi = __builtin_bit_cast(float, b);
}
USP_DEV float xhalf_max(float x) {
  float lo, hi;
  xhalf_pair(x, lo, hi);
  return fmaxf(lo, hi);
}
USP_DEV float xhalf_sum(float x) {
  float lo, hi;
  xhalf_pair(x, lo, hi);
  return lo + hi;
}

// LDS transpose read: 16 lanes supply the 16 8-byte chunks of a [4][16] 16-bit block (chunk m =
// row m/4, cols 4*(m%4)..+3); lane i of the group receives column i (4 rows).
USP_DEV u32x2 lds_read_tr16(USP_LDS const char* p) {
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((USP_LDS s16x4*)p);
  return __builtin_bit_cast(u32x2, v);
}

// Drain this wave's LDS-DMA before the barrier that publishes a tile to the other waves.  hipcc happens to
// put an `s_waitcnt vmcnt(0)` in front of the first ds_read_tr16 after a DMA (it cannot tell the double
// buffers apart) but not in front of plain ds_read_b128, so relying on it is fragile; measured cost of the
// explicit drain: none.  (Replacing the tr16 reads by asm reads, which removes hipcc's mid-tile wait, also
// measured no gain: by then the tile has landed.)
USP_DEV void dma_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// LDS-DMA: buffer_load_dwordx4 ... lds.  Each lane fetches 16 bytes at rsrc.base + voffset; the wave's
// 64 x 16 bytes land linearly at the wave-uniform LDS address `lds_dst`.  (The builtin is unknown to
// the HOST pass, which would silently drop the enclosing kernel's stub -- hence the guard.)
template <typename Rsrc> USP_DEV void lds_dma16(Rsrc rsrc, USP_LDS char* lds_dst, int voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (USP_LDS void*)lds_dst, 16, voffset, 0, 0, 0);
#endif
}
// ... at rsrc.base + soffset + voffset, `soffset` wave-uniform (an SGPR operand of the instruction): a streaming loop
// keeps ONE descriptor and advances a scalar byte offset instead of rebuilding the descriptor per tile.
template <typename Rsrc> USP_DEV void lds_dma16(Rsrc rsrc, USP_LDS char* lds_dst, int voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (USP_LDS void*)lds_dst, 16, voffset, soffset, 0, 0);
#endif
}

// Pin a value to the accumulator (AGPR) register file at this point.
USP_DEV void pin_agpr(f32x16& acc) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+a"(acc));
#endif
}

// The value must have been computed by this point of the instruction stream (an opaque use: no instruction is
// emitted).  Keeps pure arithmetic from being sunk across basic blocks into the block of its first real use.
USP_DEV void pin_here(uint32_t& w) { asm volatile("" : "+v"(w)); }

USP_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Host side: record which kernels a flash call launches (usp_last_launch_kinds; defined in usp_elementwise.hip)
void launch_kinds_reset();
void launch_kinds_note(int kind);

// Persistent workgroups: a launch has min(items, resident workgroup slots) workgroups and each walks a
// static list of work items.  The dispatcher places workgroup id on XCD id % 8; every XCD owns a
// contiguous run of the item list (all sharers of one K/V sit behind one L2) and deals it out to its
// workgroups in passes of alternating direction, which pairs a heavy causal item with a light one.
// Saves the per-workgroup launch / drain and lets item i+1's first loads overlap item i's epilogue
// (measured: ~14 us -> ~5 us of fixed cost per 256-row forward item at C2).
struct ItemWalk {
  int wg_l, wgs_l, items_l, item0;
  USP_DEV explicit ItemWalk(int n_items) {
    const int n_wg = gridDim.x;
    const bool by_xcd = ((n_wg & 7) == 0) && ((n_items & 7) == 0);
    wg_l = by_xcd ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;        // index inside the XCD
    wgs_l = by_xcd ? (n_wg >> 3) : n_wg;
    items_l = by_xcd ? (n_items >> 3) : n_items;
    item0 = by_xcd ? (int)(blockIdx.x & 7) * items_l : 0;
  }
  // item of pass `pass`, or -1 when this workgroup's list is exhausted
  USP_DEV int at(int pass) const {
    const int loc = pass * wgs_l + ((pass & 1) ? (wgs_l - 1 - wg_l) : wg_l);
    return loc < items_l ? item0 + loc : -1;
  }
  // Fewer than 8 heads: the tiles of a head that spans several XCDs are dealt to them evenly (usp_item_deal.h).
  // `w` = item id as returned by at(); returns the id to decode (head * n_inner + tile).
  USP_DEV int dealt(int w, int n_inner) const { return usp_deal_item(w, n_inner, items_l); }
  // ... and the query heads of one KV group side by side inside a run that holds several whole heads (round 6; walk_g = G, or 1 = off)
  USP_DEV int grouped(int w, int n_inner, int walk_g) const { return usp_group_item(w, n_inner, items_l, walk_g); }
};

// Dynamic item queue, used for packed batches (sequences of unequal length defeat any static split: the
// item list is sized by the LONGEST sequence, most items of short sequences are empty, and the heavy
// ones cluster).  Items are (bh, t): bh = batch*heads + head, t = tile along the owned sequence, t = 0 the
// heaviest.  XCD x (= workgroup id % 8) serves the heads bh = x (mod 8), so all sharers of one K/V still
// sit behind one L2; its workgroups pull (t, bh) pairs t-major from an atomic counter, i.e. heaviest
// first (dynamic LPT), skip empty items without leaving the fetch, and help the other XCDs' lists once
// their own is drained.  sched[0..7] = counters, sched[8] = finished workgroups; the last workgroup to
// finish zeroes the block again, so the caller zeroes it only once.  Launch shapes: resident workgroups
// that pull until the queue is dry (default), or -- USP_LAUNCH_INTERLEAVE -- one workgroup per potential
// item that pulls ONE item and exits, so that other streams' kernels can become resident in between.  Thread 0 fetches and broadcasts
// through two alternating LDS slots (one barrier per fetch).
struct ItemQueue {
  int* sched;              // device control block (16 ints)
  const int* seq;          // (first_row, rows) table of the owned side
  int n_bh, n_t, heads_per_b, unit;
  int reverse;             // 1: tile index = n_t - 1 - t (late tiles are the heavy ones)
};

// Thread 0 only.  Deliberately NOT inlined: called once per item, when almost nothing is live, so the call
// costs nothing, while its divisions and loop state inlined into the flash kernels cost 40-70 spilled VGPRs.
// The queue is passed BY REFERENCE on purpose: that is the 48-byte private segment (4 scratch stores, once, at
// kernel entry, none in any loop) the kernels report.  By value the call pins the struct in argument registers
// across the whole kernel and the flash kernels spill 37-70 VGPRs instead (tried in round 2).
// `state` = lists already drained by this workgroup; returns the item (bh * n_t + t) or -1.
__attribute__((noinline)) USP_DEV int item_queue_fetch(
    const ItemQueue& q, int& state) {
  const int x = blockIdx.x & 7;
  for (; state < 8; ++state) {
    const int xq = (x + state) & 7;
    const int nbh_x = xq < q.n_bh ? (q.n_bh - xq + 7) >> 3 : 0;
    const int total = nbh_x * q.n_t;
    while (total > 0) {
      const int j = atomicAdd(&q.sched[xq], 1);
      if (j >= total) break;
      const int t = j / nbh_x;
      const int bh = xq + 8 * (j - t * nbh_x);
      const int rows = q.seq[2 * (bh / q.heads_per_b) + 1];
      if ((q.reverse ? q.n_t - 1 - t : t) * q.unit >= rows) continue;   // empty item of a short sequence
      return bh * q.n_t + t;
    }
  }
  return -1;
}

// Thread 0, exactly once per workgroup, after its last fetch: the last workgroup to get here zeroes the block.
__attribute__((noinline)) USP_DEV void item_queue_release(const ItemQueue& q) {
  if (atomicAdd(&q.sched[8], 1) == (int)gridDim.x - 1) {        // every other workgroup has stopped fetching
    for (int i = 0; i < 9; ++i) q.sched[i] = 0;
  }
}

// all threads; `slots` = 2 ints of LDS; returns the item or -1
USP_DEV int item_queue_next(const ItemQueue& q, int& state, USP_LDS int* slots, int pass) {
  if (threadIdx.x == 0) slots[pass & 1] = item_queue_fetch(q, state);
  __syncthreads();
  return slots[pass & 1];
}

}  // namespace usp
