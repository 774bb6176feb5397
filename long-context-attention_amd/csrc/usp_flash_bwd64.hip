// Blockwise flash-attention backward for gfx950, dK/dV launch, ONE WAVE PER SIMD.  C ABI: usp_flash_bwd (include/usp_hip.h);
// replaces -- together with the dQ launch of usp_flash_bwd.hip -- the reference's `bwd-only` block kernel
// (yunchang/kernels/attention.py:205-250).
//
// The 8-wave dK/dV kernel of usp_flash_bwd.hip pairs two 32-key waves per SIMD (role A: S, P, dV; role B: dP, dS, dK) and
// is bound by the LDS fragment reads every MFMA drags along (1.5 per MFMA and wave: -13 % without them,
// profiles/r03_bwd_ablations.txt).  Here a wave owns 64 keys and its SIMD's whole register file:
//   workgroup = 4 waves = 128 keys of one (batch, kv head[, query head]); waves 0,1: role A for key slices 0,1 (64 keys
//   each), waves 2,3: role B for the same slices, one tile behind A (P reaches B through LDS, ordered by the per-tile
//   barrier that exists anyway);
//   a[0:127]   dV^T (A) / dK^T (B) accumulators, 2 key blocks x 4 dim tiles       (asm MFMAs, "+a")
//   a[128:191] K (A) / V (B) fragments of the wave's 64 keys: B operand of the S / dP chains, never copied
//   v[...]     S / dP of the two 32-row halves of the streamed tile x 2 key blocks (64), packed P / dS (32), fragments
// so that every Q / dO fragment read from LDS serves TWO MFMAs (0.75 reads per MFMA) and the element work is 2.5 VALU per
// MFMA.  A tile is 64 MFMA slots per wave:  chain(h0) | chain(h1) | grad(h0) | grad(h1)  (16 each); the 64 elements
// (P = exp2(S c - lse) for A, dS = P (dP - delta) for B, + the 16-bit packs) stream through slots 16 .. 53 in the order the
// gradient MFMAs need them, the tile's 8 LDS-DMA pieces (Q / dO tile two iterations ahead of B) through slots 0 .. 15.
// MFMAs are inline asm (see usp_mfma64.hpp for why and for the hazards hipcc cannot see); tools/mfma_hazards.py checks
// the emitted stream.
#include <stdlib.h>

#include <type_traits>

#include "usp_bwd_params.hpp"
#include "usp_common.hpp"
#include "usp_hip.h"
#include "usp_mfma64.hpp"

namespace usp {

#ifndef USP_B64_STATW
#define USP_B64_STATW 0
#endif
constexpr int kB64_STATW = USP_B64_STATW;    // the wave that stages a tile's statistics (a role-A wave: see stat_wave below)
#ifndef USP_B64_DMA_PH
#define USP_B64_DMA_PH 0
#endif
constexpr int kB64_DMA_PH = USP_B64_DMA_PH;   // phase whose slots carry the next tile's DMA pieces (0: the first chain)
// (round 5: the element stream opens at slot 20 instead of 16 -- swept at the N = 1 workload's shape on two boxes, -0.7 ... -1.5 %
// on this kernel, -0.7 % on a 16K group launch, C2 inside the noise; profiles/r05_slot_sweep.txt.  The -D overrides exist for such sweeps only.)
#ifndef USP_B64_E0
#define USP_B64_E0 20
#endif
constexpr int kB64_E0 = USP_B64_E0;      // first and one-past-last slot of the element stream (64 elements; chain(h0) ends at 16, the
#ifndef USP_B64_E1
#define USP_B64_E1 54
#endif
constexpr int kB64_E1 = USP_B64_E1;      // gradient MFMAs of k-step (h, k2) start at 32 + 16 h + 8 k2)

// dev build -DUSP_B64_TIMING: where an iteration's time goes (s_memtime stamps summed per wave, printed for a few waves;
// profiles/r04_run21..23*.log)
#ifdef USP_B64_TIMING
#define USP_TM(...) __VA_ARGS__
#else
#define USP_TM(...)
#endif

template <int DT, bool CAUSAL>
__global__ __launch_bounds__(256, 1) void flash_bwd_dkdv64_kernel(const BwdParams /* read through the kernarg segment */) {
  using E = Elem<DT>;
  using M = M64<DT>;
  constexpr int D = 128, OWN = 128;
  constexpr int ROWB = D * 2;
  constexpr int TILEB = kTile * ROWB;            // one streamed matrix tile (64 rows)
  constexpr int STATB = 2 * kTile * 4;           // lse2 + (-delta) of the tile's rows
  constexpr int BUFB = 2 * TILEB + STATB;
  constexpr int NBUF = 3;
  constexpr int PSLOT = 8192;                    // P of one 64 x 64 block, 16-bit
  constexpr int POFF = NBUF * BUFB;              // P exchange: [2 slices][2 slots][PSLOT]
  constexpr int NKT = D / 16, NDJ = D / 32;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // The dynamic LDS block is the kernel's only LDS object: it starts at LDS address 0.  Addresses are formed from that
  // integer, not from the symbol -- hipcc does not fold the symbol's value and spends a v_add (of 0) per address on it.
  if ((uint32_t)(uintptr_t)(USP_LDS char*)smem_raw != 0u) __builtin_trap();
  USP_LDS char* smem = (USP_LDS char*)(uintptr_t)0;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 1;                    // 0: A (S, P, dV)   1: B (dP, dS, dK)
  const int slice = wave & 1;
  const int l31 = lane & 31;
  const int hi = lane >> 5;
  // the argument block stays in the kernarg segment (see usp_flash_fwd64.hip: held in SGPRs it fills the scalar file)
  typedef const __attribute__((address_space(4))) BwdParams* KArgs;
  KArgs p = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));

  // ---- lane-constant addresses ----------------------------------------------------------------------------------------
  // LDS-DMA: a tile (64 rows of Q, 64 rows of dO) is 8 GROUPS of 16 rows, a group 4 pieces of 4 rows (1 KiB, one wave
  // instruction); one M0 write per group, the pieces by immediate offset (usp_mfma64.hpp).  The slot swizzle of row
  // 16g + 4i + l/16 is ((l/16) << 2) | i: piece i's per-lane offset is piece 0's with 16*i XORed in.  The groups are dealt
  // UNEVENLY: a role-A wave (whose stream carries the 64 exponentials and is the longer one) stages one group of Q, a
  // role-B wave one group of Q and two of dO -- a piece costs its wave about 35 cycles (profiles/r04_run18*).
  const int dma_row = lane >> 4;                 // (+ 16 * group + 4 * piece rows, through the scalar offset)
  const int dma_c8 = ((lane & 15) ^ ((lane >> 4) << 2)) * 16;
  const int q_voff = dma_row * (int)p->q_ss * 2 + dma_c8, do_voff = dma_row * (int)p->do_ss * 2 + dma_c8;
  // row read (A operand of the S / dP chain): tile row 32h + l31, logical slot 2t + hi; the swizzle does not depend on h
  const int rd_base = l31 * ROWB + ((hi ^ tile_swz<D>(l31)) * 16);            // ^ (32 t), + h * 32 * ROWB
  // transpose read (A operand of the gradient MFMAs) for dim tile dj, element half e, k-step ks: the 16-lane group reads
  // the [4 rows][16 dims] block rows 16ks + 8e + 4hi + (0..3), dims 32dj + 16*grp + (0..15); lane i supplies row i>>2,
  // dims 4*(i&3)..+3
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
  USP_LDS char* pex = smem + POFF + slice * 2 * PSLOT + lane * 16;   // + slot*PSLOT + ((2h + kb)*2 + k2)*1024

  const ItemWalk walk(p->n_items);               // persistent workgroups (usp_common.hpp)
  for (int pass = 0;; ++pass) {
  int w = walk.at(pass);
  if (w < 0) break;
  asm volatile("" : "+s"(p));
  USP_TM(const uint64_t tm_item = __builtin_amdgcn_s_memtime();)
  w = walk.dealt(w, p->nblk);
  const int blk = w % p->nblk;                   // early key blocks are seen by most rows: first
  int rest = w / p->nblk;
  int g = 0, cut = 0;                            // g: which run of `gsub` query heads of the KV group this item streams
  if (p->qsplit > 1) { cut = rest % p->qsplit; rest /= p->qsplit; }
  if (p->ngrp > 1) { g = rest % p->ngrp; rest /= p->ngrp; }
  const int hkv = rest % p->Hkv, b = rest / p->Hkv;
  const int gsub = p->gsub;
  const int h0 = hkv * p->G + g * gsub;
  const int own0 = blk * OWN;
  const int off = p->causal_off;
  const int ow = own0 + slice * 64;              // first key of this wave

  // K (role A) or V (role B) fragments of this wave's 64 keys, loaded straight into the accumulator file.  The loads are
  // issued from asm and waited for here: with loads hipcc can see, it still counts them as pending at the headers of the
  // streaming loops and puts a cascade of s_waitcnt vmcnt(20 .. 0) in front of the fragments' first use in EVERY iteration
  // -- its vmcnt(0) then waits for the statistics wave's fresh loads, a memory round trip per tile.
  u32x4 rf[2][NKT];
  {
    const char* pr[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int orow = ow + 32 * kb + l31;
      const int orow_c = orow < p->Sk ? orow : p->Sk - 1;
      pr[kb] = (role == 0 ? p->k + 2 * (b * p->k_sb + (int64_t)orow_c * p->k_ss + hkv * p->k_sh)
                          : p->v + 2 * (b * p->v_sb + (int64_t)orow_c * p->v_ss + hkv * p->v_sh)) + 16 * hi;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // (both key blocks' 16 loads, ONE wait: one memory round trip per item instead of two)
    asm volatile("global_load_dwordx4 %0, %16, off\n\tglobal_load_dwordx4 %1, %16, off offset:32\n\t"
                 "global_load_dwordx4 %2, %16, off offset:64\n\tglobal_load_dwordx4 %3, %16, off offset:96\n\t"
                 "global_load_dwordx4 %4, %16, off offset:128\n\tglobal_load_dwordx4 %5, %16, off offset:160\n\t"
                 "global_load_dwordx4 %6, %16, off offset:192\n\tglobal_load_dwordx4 %7, %16, off offset:224\n\t"
                 "global_load_dwordx4 %8, %17, off\n\tglobal_load_dwordx4 %9, %17, off offset:32\n\t"
                 "global_load_dwordx4 %10, %17, off offset:64\n\tglobal_load_dwordx4 %11, %17, off offset:96\n\t"
                 "global_load_dwordx4 %12, %17, off offset:128\n\tglobal_load_dwordx4 %13, %17, off offset:160\n\t"
                 "global_load_dwordx4 %14, %17, off offset:192\n\tglobal_load_dwordx4 %15, %17, off offset:224\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&a"(rf[0][0]), "=&a"(rf[0][1]), "=&a"(rf[0][2]), "=&a"(rf[0][3]), "=&a"(rf[0][4]), "=&a"(rf[0][5]),
                   "=&a"(rf[0][6]), "=&a"(rf[0][7]), "=&a"(rf[1][0]), "=&a"(rf[1][1]), "=&a"(rf[1][2]), "=&a"(rf[1][3]),
                   "=&a"(rf[1][4]), "=&a"(rf[1][5]), "=&a"(rf[1][6]), "=&a"(rf[1][7])
                 : "v"(pr[0]), "v"(pr[1]) : "memory");
#endif
  }

  // Role A's K fragments are pre-multiplied by scale * log2(e) (rounded to the 16-bit type once per item), so that the S
  // chain, started from -lse * log2(e), ends in the exponent itself: P = exp2(chain) with no per-element multiply-add.
  // Role B runs the same code with factor 1 (exact): no branch around registers the asm statements own.
  {
    const float kf = role == 0 ? p->scale_log2 : 1.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
        u32x4 x = rf[kb][t];
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = E::pack2(E::lo(x[e]) * kf, E::hi(x[e]) * kf);
        rf[kb][t] = x;
        pin_agpr4(rf[kb][t]);
      }
  }

  int t_begin = 0, t_end = (p->Sq + kTile - 1) / kTile;
  if (CAUSAL) {
    const int first_q = own0 - off > 0 ? own0 - off : 0;
    t_begin = first_q / kTile;
    if (t_begin > t_end) t_begin = t_end;
  }
  if (p->qsplit > 1) {                           // this item's cut of the query tiles [t_begin, t_end): equal runs
    const int per = (t_end - t_begin + p->qsplit - 1) / p->qsplit;
    t_begin = t_begin + cut * per < t_end ? t_begin + cut * per : t_end;
    t_end = t_begin + per < t_end ? t_begin + per : t_end;
  }
  // The item streams the tiles [t_begin, t_end) of `gsub` query heads of its KV group, one head behind the other, into
  // the same accumulators (round 6: the GQA loop inside the workgroup -- K / V fragments, their pre-scale and the epilogue
  // once per KV head instead of once per query head, and with gsub == G no fp32 per-head partials and no reduce launch).
  // Head by head: the two-stage pipeline (role B one tile behind role A, the DMA one tile ahead) drains and refills between
  // heads -- about two tile times per head, against >= 64 tiles -- so that the streaming loops carry no head state at all:
  // a seamless stream (the cursor re-based inside the loop by a uniform branch) cost every loop 40 - 140 instructions per
  // tile in SGPR spills and accumulator copies (tools/r05/census.sh).
  const int n_iter = t_end - t_begin;             // tiles per head

  // ---- LDS-DMA staging of the Q / dO tiles: running cursors (base pointer + remaining bytes, advanced per tile) -------
  const int64_t tb1 = (int64_t)kTile * p->q_ss * 2, tb2 = (int64_t)kTile * p->do_ss * 2;     // bytes per tile step
  int q_step = 4 * (int)p->q_ss * 2 - 1024, do_step = 4 * (int)p->do_ss * 2 - 1024;
  // groups of this wave: A wave a: Q group a;  B wave b: Q group 2 + b, dO groups 2b and 2b + 1
  const int q_rowb = (int)p->q_ss * 2, do_rowb = (int)p->do_ss * 2;
  const int gq = role == 0 ? slice : 2 + slice, gd = 2 * slice;
  int lds_q = gq * 4096, lds_d = TILEB + gd * 4096;                            // LDS offsets of the groups inside a buffer
  const char *q_cur = nullptr, *do_cur = nullptr;
  int rows_q = 0, rows_d = 0;                    // valid rows from the cursors on (<= 0: nothing left, lanes read 0)

  const float *lse_h = nullptr, *dl_h = nullptr;                               // row statistics of the item's head
  int st_row = 0;                                                             // first row of the cursor's tile
  float st_lse = 0.f, st_delta = 0.f;
  bool st_in = false;
  const bool stat_wave = wave == kB64_STATW;  // the wave that stages the tile's statistics (a role-A wave: the branch around its loads breaks role B's register allocation)
  auto pf_head = [&](int h) {                    // base the cursors on query head h, tile t_begin (scalar work only)
    // (the cursors point at this wave's groups: 16 gq / 16 gd rows into the tile)
    q_cur = p->q + 2 * (b * p->q_sb + h * p->q_sh) + t_begin * tb1 + (int64_t)gq * 16 * q_rowb;
    do_cur = p->dout + 2 * (b * p->do_sb + h * p->do_sh) + t_begin * tb2 + (int64_t)gd * 16 * do_rowb;
    rows_q = p->Sq - t_begin * kTile - 16 * gq;
    rows_d = p->Sq - t_begin * kTile - 16 * gd;
    lse_h = p->lse + b * p->lse_sb + h * p->lse_sh;
    dl_h = p->delta + b * p->dl_sb + h * p->dl_sh;
    st_row = t_begin * kTile;
  };
  u32x4 q_rs, do_rs;
  int dma_buf = 0;
  // open the cursor's tile for LDS buffer `buf` (descriptors) and advance the cursor: scalar work only, no branch -- it
  // runs inside the MFMA stream.  EVERY memory operation of the loop is issued from asm: a load hipcc can see makes it
  // guard the LDS reads that follow with vmcnt waits, which drain the DMA queue in the middle of the tile.
  auto dma_open = [&](int buf) {
    q_rs = make_rsrc_rows(q_cur, rows_q, 16, q_rowb, 2 * D);
    do_rs = make_rsrc_rows(do_cur, rows_d, 32, do_rowb, 2 * D);
    dma_buf = buf;
    q_cur += tb1;
    do_cur += tb2;
    rows_q -= kTile;
    rows_d -= kTile;
  };
  // the statistics wave fetches the 2 x 64 row statistics of the cursor's tile (raw, one row per lane) in front of the
  // stream ...
  auto stats_fetch = [&]() {
    if (stat_wave) {
      const int r = st_row + lane;
      st_in = r < p->Sq;                         // rows past the end: P = 0 (their Q / dO rows read as zero)
      const int rc = st_in ? r : p->Sq - 1;
      const float* pl = lse_h + rc;
      const float* pd = dl_h + rc;
      asm volatile("global_load_dword %0, %2, off\n\tglobal_load_dword %1, %3, off" : "=&v"(st_lse), "=&v"(st_delta) : "v"(pl), "v"(pd) : "memory");
    }
    st_row += kTile;
  };
  // piece n of the opened tile: n < 4 -> Q piece n, else dO piece n - 4
  auto dma_piece = [&](int n) {                  // n < 4: the Q group; 4 .. 7 / 8 .. 11: the first / second dO group (role B)
    asm volatile("" : "+s"(lds_q), "+s"(lds_d), "+s"(q_step), "+s"(do_step));
    const int i = n & 3;
    if (n < 4) lds_dma16_asm(q_rs, lds_q + dma_buf * BUFB, q_voff ^ (16 * i), i * q_step, i);
    else if (n < 8) lds_dma16_asm(do_rs, lds_d + dma_buf * BUFB, do_voff ^ (16 * i), i * do_step, i);
    else lds_dma16_asm(do_rs, lds_d + 4096 + dma_buf * BUFB, do_voff ^ (16 * i), 4096 + 4 * do_step + i * do_step, i);
  };
  // ... and behind the wave's vmcnt(0) at the end of the iteration stores what the roles consume, the constants their
  // chains START from (the C operand of a chain's first MFMA): -lse * log2(e) for role A (-inf for a row without visible
  // keys: P = 0) and -delta for role B
  auto stats_store = [&](int buf) {
    if (stat_wave) {
      asm volatile("" : "+v"(st_lse), "+v"(st_delta));         // (written by the asm loads above, complete behind dma_drain)
      const float l2 = (st_in && st_lse != USP_NEG_INF) ? -st_lse * kLog2e : -__builtin_inff();
      *(USP_LDS float*)(smem + buf * BUFB + 2 * TILEB + 4 * lane) = l2;
      *(USP_LDS float*)(smem + buf * BUFB + 2 * TILEB + 4 * kTile + 4 * lane) = st_in ? -st_delta : 0.f;
    }
  };

  f32x16 acc[2][NDJ];                            // dV^T (role A) / dK^T (role B): [key block][dim tile]
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int dj = 0; dj < NDJ; ++dj) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[kb][dj][r] = 0.f;
      pin_agpr(acc[kb][dj]);
    }

  // The streaming loop is instantiated per role, with ROLE a compile-time constant, and the role is chosen by ONE branch
  // around the whole loop: both roles execute the same barrier sequence (n_iter + 1 barriers: role B works one tile
  // behind role A, so A idles in the last iteration and B in the first).  The tile body is STRAIGHT-LINE code: with the
  // asm MFMAs under an `if (active)` hipcc copies all 128 accumulator registers twice per tile (the "+a" ties meet a phi).
  // So nothing in the loop is conditional: a tile no row of which sees the wave's keys (the first tile of a causal range,
  // for the upper key slice) runs like any other, fully masked -- its P is 0 -- and the diagonal tiles, which are the
  // FIRST n_mask tiles of a causal item, run in a loop instance of their own (MASK) that applies the mask to S.
  int n_mask = 0;                                  // leading tiles in which some (row, key) pair of this wave is masked
  if (CAUSAL) {
    const int lim = ow + 63 - off;                 // tiles with s0 < lim need the mask
    const int tm = lim > 0 ? (lim + kTile - 1) / kTile : 0;
    n_mask = tm - t_begin < 0 ? 0 : (tm - t_begin > n_iter ? n_iter : tm - t_begin);
  }
  int tile_cur = t_begin, buf_a = 0, buf_b = NBUF - 1;     // role A's tile / LDS buffers of A's and B's tiles
USP_TM(
  uint64_t tm_body = 0, tm_drain = 0, tm_bar = 0, tm_last = __builtin_amdgcn_s_memtime();
  const uint64_t tm_loop = tm_last;
)
  // Role B works one tile behind role A: the tile it takes NEXT was published a whole iteration ago, so the operands of
  // its first chain (dO row fragments, -delta) are read in the bare slots at the END of the iteration in front -- ahead of
  // the barrier, not behind it, where a lone wave would sit out the LDS round trip with the matrix pipe idle.
  u32x4 fc0[NKT];
  f32x16 cst0;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) fc0[kt] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int r = 0; r < 16; ++r) cst0[r] = 0.f;
  // one iteration: [prefetch the next tile] [the tile body of this role] [publish]
  auto step = [&](auto role_c, auto mask_c, int it, bool work) __attribute__((always_inline)) {
    constexpr int ROLE = decltype(role_c)::value;
    constexpr bool MASK = decltype(mask_c)::value;
    constexpr int E0 = kB64_E0, E1 = kB64_E1;
    static_assert(E0 >= 16 && E1 <= 56 && E1 > E0, "");
    // the next tile is fetched unconditionally (past the range of the item it is a tile nobody reads; past the end of
    // the tensor its descriptor is empty): no branch around the pieces
    const int buf_n = buf_a + 1 == NBUF ? 0 : buf_a + 1;      // (it + 1) % NBUF
    stats_fetch();
    // role B's next tile is role A's current one (buffer buf_a)
    int krn = rd_base + buf_a * BUFB + TILEB;
    asm volatile("" : "+v"(krn));
    auto next_c = [&](int kt) { fc0[kt] = *(USP_LDS const u32x4*)(smem + (krn ^ (32 * kt))); };
    auto next_s = [&](int j) {
      const f32x4 t = *(USP_LDS const f32x4*)(smem + buf_a * BUFB + 2 * TILEB + 4 * kTile + 16 * hi + 32 * j);
#pragma unroll
      for (int e = 0; e < 4; ++e) cst0[4 * j + e] = t[e];
    };
    if (work) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)        // the resident fragments STAY in the accumulator file (hipcc otherwise gives some
#pragma unroll
        for (int t = 0; t < NKT; ++t) pin_agpr4(rf[kb][t]);   // of them VGPR homes and copies them in front of every MFMA)
      const int my_it = it - ROLE;
      const int buf = ROLE == 0 ? buf_a : buf_b;
      const int s0 = (ROLE == 0 ? tile_cur : tile_cur - 1) * kTile;
      USP_LDS const char* x1 = smem + buf * BUFB;              // Q tile
      USP_LDS const char* x2 = x1 + TILEB;                     // dO tile
      USP_LDS const char* xg = ROLE == 0 ? x2 : x1;            // transpose-read operand of the gradient
      USP_LDS const char* stat = x1 + 2 * TILEB + (ROLE == 0 ? 0 : 4 * kTile) + 16 * hi;
      USP_LDS char* pslot = pex + (my_it & 1) * PSLOT;
      f32x16 sc[2][2];                                         // S (A) / dP - delta (B): [32-row half][key block]
      u32x4 pk[2][2][2];                                       // packed P (A) / dS (B): [half][key block][k-step]
      u32x4 pin[2][2][2];                                      // role B: P received from A
      f32x16 cst[2];                                           // the chains' start constants of the two halves' rows
      // the buffer base goes INTO the swizzled offset before the XOR (bases are multiples of 256, the XOR touches bits 5-7:
      // (kr ^ 32t) + base == (kr + base) ^ 32t), so a fragment address is one v_xor, not a v_xor and a v_add
      int kr = rd_base + buf * BUFB + (ROLE == 0 ? 0 : TILEB);
      asm volatile("" : "+v"(kr));       // opaque per tile: hipcc otherwise hoists the eight kr ^ 32t and keeps them live
      auto load_stats = [&](int h) {             // register r of a 32x32 C tile is row (r & 3) + 8 (r >> 2) + 4 hi
        if (ROLE == 1 && h == 0) { cst[0] = cst0; return; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 t = *(USP_LDS const f32x4*)(stat + 128 * h + 32 * j);
#pragma unroll
          for (int e = 0; e < 4; ++e) cst[h][4 * j + e] = t[e];
        }
      };
      // element n of half h, in the order the gradient k-steps need them: n = 16*k2 + 8*kb + r8 -> sc[h][kb][8*k2 + r8]
      // role A: P = exp2(S*c - lse2); role B: dS = P * (dP - delta) -- the dP chain STARTS from -delta (its C operand)
      auto elem = [&](int h, int n) {
        const int k2 = n >> 4, kb = (n >> 3) & 1, r = 8 * k2 + (n & 7);
        float val;
        if (ROLE == 0) {
          val = fast_exp2(sc[h][kb][r]);       // the chain computed (K * scale * log2 e) . q - lse * log2 e
        } else {
          const uint32_t wd = pin[h][kb][k2][(r & 7) >> 1];
          val = ((r & 1) ? E::hi(wd) : E::lo(wd)) * sc[h][kb][r];
        }
        sc[h][kb][r] = val;
      };
      auto elem_pack = [&](int h, int n) {                      // the odd element of a pair packs it
        const int k2 = n >> 4, kb = (n >> 3) & 1, r = 8 * k2 + (n & 7);
        if (r & 1) pk[h][kb][k2][(r & 7) >> 1] = E::pack2(sc[h][kb][r - 1], sc[h][kb][r]);
        if (ROLE == 0 && (r & 7) == 7)                          // 8 elements done: hand one k-step of P to B
          *(USP_LDS u32x4*)(pslot + ((2 * h + kb) * 2 + k2) * 1024) = pk[h][kb][k2];
      };
      // the element stream of the tile: 64 elements (half 0, then half 1) over slots [E0, E1).  Role A packs LAG elements
      // behind the exponentials: a v_cvt_pk straight behind the v_exp that feeds it costs a wait state (trans -> VALU),
      // which hipcc pads with an s_nop -- an issue slot per pair in the densest part of the stream.
      constexpr int LAG = ROLE == 0 ? 2 : 0;
      auto elem_slot = [&](int sl) {
        if (sl < E0 || sl >= E1) return;
#pragma unroll
        for (int n = (sl - E0) * 64 / (E1 - E0); n < (sl - E0 + 1) * 64 / (E1 - E0); ++n) {
          elem(n >> 5, n & 31);
          if (n >= LAG) elem_pack((n - LAG) >> 5, (n - LAG) & 31);
        }
        if (sl == E1 - 1) {
#pragma unroll
          for (int n = 64 - LAG; n < 64; ++n) elem_pack(n >> 5, n & 31);
        }
      };
      auto apply_mask = [&](int h) {                            // role A only: query row i sees key j iff j <= i + off
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const int d = ow + 32 * kb + l31 - off - s0 - 4 * hi - 32 * h;     // one VGPR; thresholds are inline constants
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (d > (r & 3) + 8 * (r >> 2)) sc[h][kb][r] = USP_NEG_INF;
        }
      };
      // LDS fragments are read ONE PHASE AHEAD of the MFMAs that take them (a lone wave cannot hide an LDS round trip
      // behind anything: every s_waitcnt that has to wait idles the matrix pipe): the slots of chain(h0) carry the reads
      // of chain(h1), those of chain(h1) the reads of grad(h0), those of grad(h0) the reads of grad(h1); only chain(h0)'s
      // own fragments are read in one burst behind the barrier that published the tile.
      u32x4 fc[2][NKT];                                        // row-read fragments of the two halves' chains
      u32x4 xa[2][2 * NDJ];                                    // transpose-read fragments of the two halves' gradients
      auto rd_c = [&](int h, int kt) { fc[h][kt] = *(USP_LDS const u32x4*)(smem + h * 32 * ROWB + (kr ^ (32 * kt))); };
      auto rd_g = [&](int h, int f) {                          // fragment f = NDJ*k2 + dj
        USP_LDS const char* xb = xg + (2 * h + f / NDJ) * 16 * ROWB;
        const u32x2 a0 = lds_read_tr16(xb + tr_addr[f % NDJ][0]);
        const u32x2 a1 = lds_read_tr16(xb + tr_addr[f % NDJ][1]);
        xa[h][f] = u32x4{a0[0], a0[1], a1[0], a1[1]};
      };
      auto chain_init = [&](int h) {                           // statistics; role B: the chains' C operand and A's P
        load_stats(h);
        if (ROLE == 1) {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)                        // fetch A's P of this half early
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) pin[h][kb][k2] = *(USP_LDS const u32x4*)(pslot + ((2 * h + kb) * 2 + k2) * 1024);
        }
      };
      // ---------------- S / dP chains: slots [16h, 16h + 16) ----------------
      auto chain_phase = [&](int h) {
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) {
          const int kt = sl >> 1, kb = sl & 1;
          if (kt == 0) M::template s_first_c<MASK>(sc[h][kb], fc[h][kt], rf[kb][kt], cst[h]);
          else M::template s_next<MASK>(sc[h][kb], fc[h][kt], rf[kb][kt]);
          if (h == kB64_DMA_PH && sl == 0) {         // behind the first MFMA: its scalar work must not idle the matrix pipe
            __builtin_amdgcn_sched_barrier(0);
            dma_open(buf_n);
          }
          if (sl < NKT) { if (h == 0) rd_c(1, sl); else rd_g(0, sl); }       // the next phase's fragments: slots 0 .. 7
          if (h == 0 && sl == 8) chain_init(1);
          elem_slot(16 * h + sl);
          if (h == kB64_DMA_PH) {                    // the next tile's pieces: role A slots 1, 5, 9, 13; role B slots 1 .. 12
            if (ROLE == 0 && (sl & 3) == 1) dma_piece(sl >> 2);
            if (ROLE == 1 && sl >= 1 && sl <= 12) dma_piece(sl - 1);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      // ---------------- gradient MFMAs: slots [32 + 16h, 48 + 16h) ----------------
      auto grad_phase = [&](int h) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int f = i >> 1, kb = i & 1;
          M::template o_acc<MASK>(acc[kb][f % NDJ], xa[h][f], pk[h][kb][f / NDJ]);
          if (h == 0 && i < 2 * NDJ) rd_g(1, i);                              // the next phase's fragments: slots 0 .. 7
          if (ROLE == 1 && h == 1 && i >= 8) next_c(i - 8);                   // role B: the next tile's first chain
          if (ROLE == 1 && h == 1 && i >= 4 && i < 8) next_s(i - 4);
          elem_slot(32 + 16 * h + i);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      chain_init(0);
      if (ROLE == 1) {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) fc[0][kt] = fc0[kt];
      } else {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) rd_c(0, kt);
      }
      __builtin_amdgcn_sched_barrier(0);
      // The fragments of a phase were all requested in the first half of the phase in front: ONE s_waitcnt lgkmcnt(0) at the
      // phase boundary finds them landed, and hipcc then drops the counted wait it otherwise puts in front of every MFMA
      // that takes a fragment (about twenty issue slots per tile).
      chain_phase(0);
      if (MASK) { mfma_settle(sc[0]); apply_mask(0); }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      chain_phase(1);
      if (MASK) { mfma_settle(sc[1]); apply_mask(1); }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      grad_phase(0);
      __builtin_amdgcn_s_waitcnt(0xc07f);
      grad_phase(1);
    } else {
      dma_open(buf_n);
#pragma unroll
      for (int n = 0; n < (ROLE == 0 ? 4 : 12); ++n) dma_piece(n);
    }
    if (ROLE == 1 && !work) {                      // (role B's idle first iteration: the same reads, outside a stream)
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) next_c(kt);
#pragma unroll
      for (int j = 0; j < 4; ++j) next_s(j);
    }
    buf_b = buf_a;
    buf_a = buf_n;
    if (ROLE == 0 || it > 0) ++tile_cur;           // (role B enters its first tile one iteration late)
    USP_TM(const uint64_t tm0 = __builtin_amdgcn_s_memtime();)
    dma_drain();            // this wave's DMA pieces of the staged tile have landed
    USP_TM(const uint64_t tm1 = __builtin_amdgcn_s_memtime();)
    stats_store(buf_n);
    __syncthreads();
    USP_TM(const uint64_t tm2 = __builtin_amdgcn_s_memtime();
           tm_body += tm0 - tm_last; tm_drain += tm1 - tm0; tm_bar += tm2 - tm1; tm_last = tm2;)
  };
  const std::integral_constant<int, 0> rA;
  const std::integral_constant<int, 1> rB;
  const std::integral_constant<bool, false> plain;
  const std::integral_constant<bool, true> masked;
  // (the vmcnt(0) in front of every loop: hipcc reloads spilled registers -- VMEM loads -- on the way here, and what it
  // still counts as pending at a loop header it waits for at the first use INSIDE the loop, in every iteration; with the
  // statistics wave's fresh loads in flight that is a memory round trip per tile)
  // a head's first tile: cursors, descriptors, all of the tile's pieces, the barrier that publishes it (every wave is past the
  // barrier that ended the head in front: nobody reads a buffer or a P slot any more)
  auto open_head = [&](int h) {
    pf_head(h);
    stats_fetch();
    dma_open(0);
#pragma unroll
    for (int n = 0; n < 4; ++n) dma_piece(n);
    if (role == 1) {
#pragma unroll
      for (int n = 4; n < 12; ++n) dma_piece(n);
    }
    dma_drain();
    stats_store(0);
    __syncthreads();
    tile_cur = t_begin; buf_a = 0; buf_b = NBUF - 1;
  };
  if (role == 0) {
    for (int hh = 0; hh < gsub; ++hh) {            // head by head: the diagonal tiles first, then the plain ones
      open_head(h0 + hh);
      int it = 0;
      __builtin_amdgcn_s_waitcnt(0x0f70);
      for (; it < n_mask; ++it) step(rA, masked, it, true);
      __builtin_amdgcn_s_waitcnt(0x0f70);
      for (; it < n_iter; ++it) step(rA, plain, it, true);
      mfma_settle(acc);          // (whatever hipcc does with the accumulators behind the loop, it does behind these wait states)
      step(rA, plain, n_iter, false);
    }
  } else {
    for (int hh = 0; hh < gsub; ++hh) {
      open_head(h0 + hh);
      step(rB, plain, 0, false);
      __builtin_amdgcn_s_waitcnt(0x0f70);
      for (int it = 1; it <= n_iter; ++it) step(rB, plain, it, true);
      mfma_settle(acc);
    }
  }

  // ---- epilogue -----------------------------------------------------------------------------------------------------------
  mfma_settle(acc);
USP_TM(
  const uint64_t tm_epi = __builtin_amdgcn_s_memtime();
  if (pass == 0 && lane == 0 && (blockIdx.x % 61) == 0)
    printf("TM wg %3d wave %d blk %2d n_iter %3d : body %8llu drain %7llu barrier %7llu  (per iteration %5llu / %4llu / %4llu)\n",
           (int)blockIdx.x, wave, blk, n_iter, (unsigned long long)tm_body, (unsigned long long)tm_drain,
           (unsigned long long)tm_bar, (unsigned long long)(tm_body / (n_iter + 1)), (unsigned long long)(tm_drain / (n_iter + 1)),
           (unsigned long long)(tm_bar / (n_iter + 1)));
)
  asm volatile("" : "+s"(p));
  // 16-bit final output of this role's tensor, rows 16-byte aligned, nothing accumulated, no head-split partials
  const bool wide = !p->split && (p->wide16 & (role == 0 ? 4 : 2)) != 0;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int orow = ow + 32 * kb + l31;
    if (wide) {
      const int orow_c = orow < p->Sk ? orow : 0;
      char* row16 = role == 0 ? p->dv16 + 2 * (b * p->dv16_sb + (int64_t)orow_c * p->dv16_ss + hkv * p->dv16_sh)
                              : p->dk16 + 2 * (b * p->dk16_sb + (int64_t)orow_c * p->dk16_ss + hkv * p->dk16_sh);
      store_row16_wide<E, NDJ>(row16, acc[kb], role == 0 ? 1.f : p->scale, hi, orow < p->Sk);
    } else if (orow < p->Sk) {
      float* o32;
      char* o16 = nullptr;
      int accf;
      const float mul = role == 0 ? 1.f : p->scale;
      if (p->split) {                            // partial slab of (head run, cut)
        const int64_t wo = (((int64_t)(g * p->qsplit + cut) * p->ws_rows + (int64_t)b * p->Sk + orow) * p->Hkv + hkv) * D;
        o32 = (role == 0 ? p->ws_dv : p->ws_dk) + wo; accf = 0;
      } else if (role == 0) {
        o32 = p->dv + b * p->dv_sb + (int64_t)orow * p->dv_ss + hkv * p->dv_sh; accf = p->accum_dv;
        if (p->dv16) o16 = p->dv16 + 2 * (b * p->dv16_sb + (int64_t)orow * p->dv16_ss + hkv * p->dv16_sh);
      } else {
        o32 = p->dk + b * p->dk_sb + (int64_t)orow * p->dk_ss + hkv * p->dk_sh; accf = p->accum_dk;
        if (p->dk16) o16 = p->dk16 + 2 * (b * p->dk16_sb + (int64_t)orow * p->dk16_ss + hkv * p->dk16_sh);
      }
#pragma unroll
      for (int dj = 0; dj < NDJ; ++dj)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int d0 = 32 * dj + 8 * g4 + 4 * hi;
          f32x4 v = {acc[kb][dj][4 * g4] * mul, acc[kb][dj][4 * g4 + 1] * mul, acc[kb][dj][4 * g4 + 2] * mul,
                     acc[kb][dj][4 * g4 + 3] * mul};
          if (accf) v += *(const f32x4*)(o32 + d0);
          if (o16) *(u32x2*)(o16 + 2 * d0) = u32x2{E::pack2(v[0], v[1]), E::pack2(v[2], v[3])};
          else *(f32x4*)(o32 + d0) = v;
        }
    }
  }
USP_TM(
  if (pass < 3 && lane == 0 && (blockIdx.x % 61) == 0)
    printf("TI wg %3d pass %d wave %d blk %2d n_iter %3d n_mask %d : prologue %6llu loops %8llu epilogue %6llu\n", (int)blockIdx.x, pass,
           wave, blk, n_iter, n_mask, (unsigned long long)(tm_loop - tm_item), (unsigned long long)(tm_epi - tm_loop),
           (unsigned long long)(__builtin_amdgcn_s_memtime() - tm_epi));
)
  }  // next item
}

bool dkdv64_serves(const BwdParams& p_in, int dtype) {
  if (p_in.seq_q || p_in.seq_k || p_in.sched || p_in.win_on) return false;
  // (fp16: round 4's first build of that instantiation copied accumulators between the register files inside the loop; with the
  // chains started from the row constants it compiles like the bf16 one -- tools/mfma_hazards.py: 0 -- and is served here too)
  // role A keeps K * scale * log2(e) in the 16-bit type: with fp16 an unusually large softmax scale could overflow it
  if (dtype != USP_BF16 && !(p_in.scale_log2 <= 8.f)) return false;
  // the pieces' swizzle is XORed into the per-lane byte offset (row part a multiple of 256 bytes); per-lane offsets and the
  // pieces' scalar offsets are 32-bit: 64 rows of Q / dO must span less than 2^31 bytes.  (Base pointer and remaining
  // bytes are 64-bit: no sequence length is refused -- the 8-wave kernel addresses a head by a 32-bit offset and is.)
  if ((p_in.q_ss * 2) % 256 != 0 || (p_in.do_ss * 2) % 256 != 0 || p_in.q_ss * 128 >= (1LL << 31) || p_in.do_ss * 128 >= (1LL << 31))
    return false;
  return true;
}

bool launch_dkdv64(const BwdParams& p_in, int dtype, bool causal, hipStream_t st, int* rc) {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  if (!dkdv64_serves(p_in, dtype)) return false;
  BwdParams p = p_in;
  p.wide16 = ((p.dk16 && !p.accum_dk && rows16_aligned(p.dk16, p.dk16_sb, p.dk16_ss, p.dk16_sh)) ? 2 : 0) |
             ((p.dv16 && !p.accum_dv && rows16_aligned(p.dv16, p.dv16_sb, p.dv16_ss, p.dv16_sh)) ? 4 : 0);
  p.nblk = (p.Sk + 127) / 128;
  p.n_items = p.B * p.Hkv * p.nblk * p.ngrp * p.qsplit;
  // persistent: one workgroup per CU; USP_LAUNCH_INTERLEAVE: one workgroup per item (the same kernel: a workgroup's item
  // list then has one entry)
  const int grid = (!p.interleave && p.n_items > cus) ? cus : p.n_items;
  constexpr size_t lds = 3 * (2 * kTile * 128 * 2 + 2 * kTile * 4) + 2 * 2 * 8192;
  if (dtype == USP_BF16) {
    if (causal) hipLaunchKernelGGL((flash_bwd_dkdv64_kernel<0, true>), dim3(grid), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((flash_bwd_dkdv64_kernel<0, false>), dim3(grid), dim3(256), lds, st, p);
  } else {
    if (causal) hipLaunchKernelGGL((flash_bwd_dkdv64_kernel<1, true>), dim3(grid), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((flash_bwd_dkdv64_kernel<1, false>), dim3(grid), dim3(256), lds, st, p);
  }
  *rc = hipGetLastError() == hipSuccess ? USP_OK : USP_ELAUNCH;
  return true;
}

}  // namespace usp
