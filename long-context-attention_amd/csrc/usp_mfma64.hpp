// One-wave-per-SIMD building blocks shared by usp_flash_fwd64.hip and usp_flash_bwd64.hip: inline-asm MFMA forms that let
// ONE kernel keep accumulators and resident operands in the accumulator half of the register file (AGPRs) while the score
// tiles live in arch VGPRs, the wait-state guards hipcc cannot place around asm MFMAs, and LDS-DMA issued from asm.
#pragma once
#include "usp_common.hpp"

namespace usp {

// ---- asm MFMA forms ---------------------------------------------------------------------------------------------------
template <int DT> struct M64;
#if defined(__HIP_DEVICE_COMPILE__)   // "a" means eax to the host pass, which then drops the kernel stubs
#define USP_M64_BODY(MN)                                                                                              \
  /* first MFMA of a score chain: C = 0; D in arch VGPRs; A = K fragment (VGPR), B = Q fragment (AGPR) */            \
  /* SAFE: two wait states in front ("VALU / v_accvgpr_write -> MFMA operand") for the code outside the steady state, */ \
  /* where hipcc may re-materialise an operand right in front of the statement                                     */ \
  template <bool SAFE = false> static USP_DEV void s_first(f32x16& s, const u32x4& a, const u32x4& q) {               \
    if (SAFE) asm volatile("s_nop 1\n\t" MN " %0, %1, %2, 0" : "=&v"(s) : "v"(a), "a"(q));                          \
    else asm volatile(MN " %0, %1, %2, 0" : "=&v"(s) : "v"(a), "a"(q));                                               \
  }                                                                                                                   \
  /* first MFMA of a chain that starts from a per-row constant: C = c (16 arch VGPRs of their own), D = s */          \
  template <bool SAFE = false> static USP_DEV void s_first_c(f32x16& s, const u32x4& a, const u32x4& q, const f32x16& c) { \
    if (SAFE) asm volatile("s_nop 1\n\t" MN " %0, %1, %2, %3" : "=&v"(s) : "v"(a), "a"(q), "v"(c));                \
    else asm volatile(MN " %0, %1, %2, %3" : "=&v"(s) : "v"(a), "a"(q), "v"(c));                                      \
  }                                                                                                                   \
  template <bool SAFE = false> static USP_DEV void s_next(f32x16& s, const u32x4& a, const u32x4& q) {                \
    if (SAFE) asm volatile("s_nop 1\n\t" MN " %0, %1, %2, %0" : "+v"(s) : "v"(a), "a"(q));                          \
    else asm volatile(MN " %0, %1, %2, %0" : "+v"(s) : "v"(a), "a"(q));                                               \
  }                                                                                                                   \
  /* O^T accumulate: C/D in AGPRs; A = V^T fragment, B = packed P (VGPRs) */                                           \
  template <bool SAFE = false> static USP_DEV void o_acc(f32x16& o, const u32x4& a, const u32x4& b) {                 \
    if (SAFE) asm volatile("s_nop 1\n\t" MN " %0, %1, %2, %0" : "+a"(o) : "v"(a), "v"(b));                          \
    else asm volatile(MN " %0, %1, %2, %0" : "+a"(o) : "v"(a), "v"(b));                                               \
  }
#else
#define USP_M64_BODY(MN)                                                                                              \
  template <bool SAFE = false> static USP_DEV void s_first(f32x16&, const u32x4&, const u32x4&) {}                    \
  template <bool SAFE = false> static USP_DEV void s_first_c(f32x16&, const u32x4&, const u32x4&, const f32x16&) {}   \
  template <bool SAFE = false> static USP_DEV void s_next(f32x16&, const u32x4&, const u32x4&) {}                     \
  template <bool SAFE = false> static USP_DEV void o_acc(f32x16&, const u32x4&, const u32x4&) {}
#endif
template <> struct M64<0> { USP_M64_BODY("v_mfma_f32_32x32x16_bf16") };
template <> struct M64<1> { USP_M64_BODY("v_mfma_f32_32x32x16_f16") };

USP_DEV void pin_agpr4(u32x4& x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+a"(x));
#endif
}
// `n` wait states that hipcc cannot move the readers of the accumulators across (it does not know that the asm
// statements in front of it are MFMAs: "XDL write -> VALU / v_accvgpr read" needs 12 states for an 8-pass MFMA).
USP_DEV void mfma_settle(f32x16 (&o)[2][4]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 15" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[0][2]), "+a"(o[0][3]),
                            "+a"(o[1][0]), "+a"(o[1][1]), "+a"(o[1][2]), "+a"(o[1][3]));
#endif
}
USP_DEV void mfma_settle(f32x16 (&s)[2][2]) {
  asm volatile("s_nop 15" : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[1][0]), "+v"(s[1][1]));
}
USP_DEV void mfma_settle(f32x16 (&s)[2]) { asm volatile("s_nop 15" : "+v"(s[0]), "+v"(s[1])); }
// VALU write (v_cvt_pk / v_accvgpr_write) -> MFMA operand read: 2 wait states, which hipcc does not pad in front of asm
USP_DEV void operand_settle() { asm volatile("s_nop 3" ::: "memory"); }

// LDS-DMA from inline asm (buffer_load_dwordx4 ... lds: 64 x 16 bytes at rsrc.base + soffset + voffset land linearly at
// the wave-uniform LDS address M0).  hipcc does not see these loads: it puts no s_waitcnt vmcnt(0) in front of the first
// ds_read_b64_tr_b16 behind a DMA (it cannot tell the double buffers apart), so the pieces can be issued anywhere in an
// iteration; the kernel drains them itself (dma_drain) in front of the barrier that publishes the tile.  M0 is written
// by piece 0 of a tile and read by pieces 1-3 (hipcc itself never touches M0 in this kernel: tools/mfma_hazards.py checks
// the .s); the s_nop covers "SALU write M0 -> LDS-DMA".  Descriptor and offsets are SALU results (no VALU -> SGPR hazard).
USP_DEV void lds_dma16_asm(const u32x4& rsrc, int lds_dst, int voffset, int soffset, int piece) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (piece == 0)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 : : "s"(lds_dst), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
  if (piece == 1) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:1024 lds" : : "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
  if (piece == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:2048 lds" : : "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
  if (piece == 3) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:3072 lds" : : "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
#endif
}
// 4-byte pieces (64 x 4 bytes per wave-instruction: one row statistic per lane)
USP_DEV void lds_dma4_asm(const u32x4& rsrc, int lds_dst, int voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
               : : "s"(lds_dst), "v"(voffset), "s"(rsrc) : "memory");
#endif
}
// Raw buffer descriptor over `bytes` bytes from `base`; `bytes` is a running 64-bit remainder (negative: nothing left,
// every lane reads 0; beyond 32 bits: the whole 4 GiB window -- a tile never reaches that far, so the clamp is exact for
// everything a piece can address).  No tensor size is refused because of 32-bit byte arithmetic.
USP_DEV u32x4 make_rsrc(const char* base, int64_t bytes) {
  const uint64_t a = (uint64_t)base;
  const uint32_t n = bytes <= 0 ? 0u : (bytes > 0xffffffffLL ? 0xffffffffu : (uint32_t)bytes);
  return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, n, 0x00020000u};
}

// HARDWARE ASSUMPTION (both descriptor forms): the raw-buffer range check of gfx950 covers voffset + the instruction offset
// + the SCALAR offset -- a piece's row step and group live in the scalar offset, and rows past the last valid one must read
// as zero.  Checked, not assumed: kbench puts 1 MiB of NaN behind every tensor it uploads, so a lane that escapes the
// clamp multiplies a NaN into a ragged shape's result (csrc/tools/kbench.cpp: kPoisonTail; the native suite is green).
// The same for a tile cursor that counts ROWS: `rows` valid rows (any sign) of `row_bytes` each remain from `base`, a tile
// addresses at most `tile_rows` of them and `used` bytes of a row.  32-bit scalar arithmetic only (five SALU operations): the
// caller guarantees tile_rows * row_bytes < 2^31.
USP_DEV u32x4 make_rsrc_rows(const char* base, int rows, int tile_rows, int row_bytes, int used) {
  const uint64_t a = (uint64_t)base;
  const int r = rows < tile_rows ? rows : tile_rows;
  const int n = r * row_bytes - (row_bytes - used);
  return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, (uint32_t)(n > 0 ? n : 0), 0x00020000u};
}

// 16-bit store of one transposed accumulator row block (O^T / dQ^T / dK^T / dV^T: a lane holds 4-dim pieces of ONE row, the
// two half-waves interleaved in 8-byte pieces): one v_permlane32_swap per dword regroups two adjacent pieces into 16
// contiguous bytes per lane, so a row block leaves in 8 dwordx4 stores instead of 16 dwordx2 -- the store tail of these
// kernels is issue-bound (per-lane stores at a row stride).  EVERY lane of the wave must call it (the swap pairs lane l with
// l + 32, which hold the same row); `valid` gates the stores only.  `row` must be 16-byte aligned.
template <class E, int NDJ>
USP_DEV void store_row16_wide(char* row, const f32x16 (&t)[NDJ], float mul, int hi, bool valid) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int dj = 0; dj < NDJ; ++dj)
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
      const int r0 = 8 * g2;
      const uint32_t ax = E::pack2(t[dj][r0] * mul, t[dj][r0 + 1] * mul);
      const uint32_t ay = E::pack2(t[dj][r0 + 2] * mul, t[dj][r0 + 3] * mul);
      const uint32_t bx = E::pack2(t[dj][r0 + 4] * mul, t[dj][r0 + 5] * mul);
      const uint32_t by = E::pack2(t[dj][r0 + 6] * mul, t[dj][r0 + 7] * mul);
      const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
      const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
      if (valid) *(u32x4*)(row + 2 * (32 * dj + 16 * g2 + 8 * hi)) = u32x4{sx[0], sy[0], sx[1], sy[1]};
    }
#endif
}

}  // namespace usp
