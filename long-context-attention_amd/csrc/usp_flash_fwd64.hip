// Blockwise flash-attention forward for gfx950, ONE WAVE PER SIMD: a workgroup is 4 waves x 64 query rows (256 rows of
// one (batch, head), KV tile = 64 keys), every wave owns its SIMD's whole 512-entry register file.
// Same C ABI entry (usp_flash_fwd, include/usp_hip.h), same LDS images, same numerics, same epilogue (fused ring LSE
// merge) as the 8-wave kernel of usp_flash_fwd.hip, which stays the kernel of the packed / windowed / small launches and of
// K splits with short cuts (round 5: long cuts run here, see SPLIT below).  Replaces the reference's `fwd-only` block kernel (yunchang/kernels/attention.py:44-136) and
// update_out_and_lse (yunchang/ring/utils.py:10-51).
//
// Why this shape (profiles/r03_bwd_ablations.txt, tools/issue_bench.hip): with two 32-row waves per SIMD every MFMA
// drags 1.5 LDS fragment reads and 4 VALU through the SIMD's issue port; the 8-wave kernel sits at 58 % MFMA-pipe
// occupancy, which is 80-90 % of what that instruction mix allows.  A wave that owns 64 rows uses every K and V fragment
// for TWO MFMAs (0.75 LDS reads per MFMA), and with the accumulators in the accumulator half of the register file
// the arithmetic VGPRs are free for two full score tiles:
//   a[0:127]   O^T accumulators, 2 query blocks x 4 dim tiles        (asm MFMAs, "+a": hipcc would give them VGPR homes)
//   a[128:191] Q fragments, B operand of S^T = K Q^T (MFMA A/B operands may be AGPRs: no copy, ever)
//   v[...]     S^T of the tile being exponentiated + S^T of the next tile (2 x 64), packed P (32), K / V fragments in
//              flight, softmax state
// The MFMAs are inline asm (hipcc selects ONE accumulator file per function for its builtin MFMAs: with the builtins the
// score chain would land in AGPRs and every score would pay a v_accvgpr_read).  hipcc pads nothing around an asm MFMA,
// so the hazards are met by construction: a chain's result is read at least 16 MFMA slots after its last MFMA in the
// pipelined loop, and behind an explicit s_nop in the unpipelined paths; operands written by VALU (packed P) are
// complete at least one MFMA slot before the MFMA that reads them.
#include <stdlib.h>

#include <type_traits>

#include "usp_common.hpp"
#include "usp_fwd_params.hpp"
#include "usp_hip.h"
#include "usp_mfma64.hpp"

namespace usp {

// Placement of the element work and of the LDS-DMA pieces among the 64 MFMA slots of a tile (tuned on the C2 shape,
// profiles/r04_dma_probes.txt; swept again at the N = 1 workload's shape in round 5 -- every variant within +-1 % of these,
// profiles/r05_slot_sweep.txt.  Constants: the -D overrides exist for such sweeps only):
#ifndef USP_F64_NEA
#define USP_F64_NEA 42
#endif
constexpr int kF64_NEA = USP_F64_NEA;    // exp elements (of 64 per lane and tile) issued in phase A; the rest opens phase B, one per slot
#ifndef USP_F64_LEAD
#define USP_F64_LEAD 2
#endif
constexpr int kF64_LEAD = USP_F64_LEAD;    // ... of which in front of the first MFMA of phase A (it waits for the first K fragments anyway)
#ifndef USP_F64_PFK
#define USP_F64_PFK 2
#endif
constexpr int kF64_PFK = USP_F64_PFK;     // K fragments read this many fragments (= 2 MFMA slots each) ahead of their first MFMA
#ifndef USP_F64_PFV
#define USP_F64_PFV 2
#endif
constexpr int kF64_PFV = USP_F64_PFV;     // likewise the V fragments
#ifndef USP_F64_MAX0
#define USP_F64_MAX0 22
#endif
constexpr int kF64_MAX0 = USP_F64_MAX0;   // first slot of phase B that carries row-max work of the next tile
#ifndef USP_F64_DMA0
#define USP_F64_DMA0 2
#endif
constexpr int kF64_DMA0 = USP_F64_DMA0;    // slot (0..63 over both phases) behind whose MFMA the first of the iteration's 8 LDS-DMA pieces
#ifndef USP_F64_DMAS
#define USP_F64_DMAS 6
#endif
constexpr int kF64_DMAS = USP_F64_DMAS;    // goes out, and the distance to the next one

// dev build -DUSP_F64_TIMING: where an item's time goes (s_memtime stamps, printed for a few waves; profiles/r04_run28*.log)
#ifdef USP_F64_TIMING
#define USP_TM(...) __VA_ARGS__
#else
#define USP_TM(...)
#endif

// SPLIT: the K-split form (usp_fwd_args.k_splits > 1) is its own instantiation with its own argument block (FwdParams +
// FwdSplit), as in usp_flash_fwd.hip: the plain kernels keep the machine code they were profiled with.  A work item is then
// one CUT of a query tile's keys, bound by rebasing the K / V cursors, the key count and the causal offset (the tile loops
// are untouched); it writes its normalised fp32 partial and its LSE to the workspace through the not-final epilogue, and
// split_merge_kernel (usp_flash_fwd.hip) combines the cuts, the running result and the 16-bit emission afterwards.
template <int DT, bool CAUSAL, bool SPLIT = false>
__global__ __launch_bounds__(256, 1) void flash_fwd64_kernel(const FwdArgsT<SPLIT> /* read through the kernarg segment */) {
  using E = Elem<DT>;
  using M = M64<DT>;
  constexpr int D = 128;
  constexpr int kBM = 256;                    // query rows per workgroup
  constexpr int ROWB = D * 2;                 // bytes per K row
  constexpr int KBYTES = kBN * ROWB;          // one K (or V) tile
  constexpr int NKT = D / 16;                 // k-steps of K Q^T
  constexpr int NDJ = D / 32;                 // 32-wide dim tiles of O^T
  constexpr int VOFF = 2 * KBYTES;            // LDS: Kbuf[0], Kbuf[1], Vbuf[0], Vbuf[1]

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  USP_LDS char* smem = (USP_LDS char*)smem_raw;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31;
  const int hi = lane >> 5;
  // The argument block (~70 dwords) is NOT kept in SGPRs: every use reads it from the kernarg segment through a pointer
  // the compiler cannot see through (re-laundered per item and in front of the epilogue), so that a field lives in a
  // register only around its uses.  Held in SGPRs for the whole persistent loop the block alone fills the scalar file,
  // and the pipelined loop then reloads its loop invariants from spill lanes (v_readlane + 5 wait states in front of
  // every LDS-DMA that takes one as its scalar offset).
  typedef const __attribute__((address_space(4))) FwdArgsT<SPLIT>* KArgs;
  KArgs p = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));

  // ---- lane-constant addresses --------------------------------------------------------------------------------------
  // LDS-DMA pieces (1 KiB per wave-instruction; tile layouts as in usp_flash_fwd.hip).  This wave fills the CONTIGUOUS chunks
  // 4*wave .. 4*wave + 3 of every tile = K rows / V keys 16*wave + 4i .. +3 for piece i: the LDS address of a piece is
  // M0 + the instruction's immediate offset, so ONE M0 write serves the four pieces of a tile (offset 0 / 1024 / 2048 /
  // 3072; the same immediate also moves the memory address and is taken out of the scalar offset again).  Writing M0 per
  // piece costs ~65 cycles each -- the s_mov waits for the previous piece to leave the memory pipeline -- which one wave
  // per SIMD cannot hide (profiles/r04_dma_probes.txt).  K: the slot swizzle key of row 16w + 4i + l/16 is 4i | l/16, so
  // piece i's per-lane offset is piece 0's with 64*i XORed in (the row part is a multiple of 256 bytes: launch_fwd64).
  const int k_voff = (16 * wave + (lane >> 4)) * (int)p->k_ss * 2 + (((lane & 15) ^ (lane >> 4)) * 16);
  const int v_voff = (16 * wave + ((lane & 15) >> 2)) * (int)p->v_ss * 2 + (32 * (lane >> 4) + 8 * (lane & 3)) * 2;
  // row read (A operand of K Q^T): tile row 32*kb + l31, logical slot 2t + hi -> physical slot (2t) ^ (hi ^ swz).  The row
  // part is a multiple of 256 and the slot part is below 256, so the eight k-steps' addresses are ONE base with 32*t XORed
  // in (one v_xor per k-step instead of eight address registers: the loop sits at the 256-VGPR limit)
  const int k_rd = l31 * ROWB + ((hi ^ (l31 & 15)) * 16);
  const int v_rd = VOFF + hi * NDJ * 256 + ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
  const float c = p->scale_log2;

  // ---- persistent workgroups: each walks a static list of (batch, head, 256-row query tile) items (ItemWalk) ------
  const ItemWalk walk(p->n_items);
  for (int pass = 0;; ++pass) {
  int w = walk.at(pass);
  if (w < 0) break;
  asm volatile("" : "+s"(p));
  int ks = 0;                                             // (split form) the cut of the query tile's keys this item is
  if constexpr (SPLIT) {
    w = walk.dealt(w, p->nq * p->ksplit);                 // the cuts of a tile are dealt like tiles
    ks = w % p->ksplit; w /= p->ksplit;
  } else {
    w = walk.grouped(walk.dealt(w, p->nq), p->nq, p->walk_g);
  }
  const int qt_r = w % p->nq;
  int rest = w / p->nq;
  const int qt = CAUSAL ? (p->nq - 1 - qt_r) : qt_r;      // heavy (late) tiles first
  const int g = rest % p->G;
  rest /= p->G;
  const int hkv = rest % p->Hkv;
  const int b = rest / p->Hkv;
  const int h = hkv * p->G + g;

  const int q0 = qt * kBM;
  const int qw = q0 + wave * 64;                          // first row of this wave
  int off = p->causal_off;
  // (split form) first key of the cut and the number of keys in it: the item sees keys [kb, kb + sk_cut) as keys [0, sk_cut)
  int kb = 0, sk_cut = 0;
  if constexpr (SPLIT) {
    int e = p->Sk;                                        // keys any row of the 256-row tile sees
    if (CAUSAL) {
      const int lim = (q0 + kBM < p->Sq ? q0 + kBM : p->Sq) + off;
      e = lim < e ? lim : e;
    }
    const int nt_all = e > 0 ? (e + kBN - 1) / kBN : 0;
    kb = (ks * nt_all / p->ksplit) * kBN;
    int ke = (ks == p->ksplit - 1) ? p->Sk : ((ks + 1) * nt_all / p->ksplit) * kBN;
    ke = ke < p->Sk ? ke : p->Sk;
    sk_cut = ke > kb ? ke - kb : 0;
    off -= kb;
  }
  auto n_keys = [&]() { if constexpr (SPLIT) return sk_cut; else return p->Sk; };

  // ---- KV range -------------------------------------------------------------------------------------------------------
  int blk_kv_end = n_keys(), wave_kv_end = n_keys();
  if (CAUSAL) {
    const int blk_last = (q0 + kBM < p->Sq ? q0 + kBM : p->Sq) - 1;
    const int wav_last = (qw + 64 < p->Sq ? qw + 64 : p->Sq) - 1;
    blk_kv_end = blk_last + off + 1 < n_keys() ? blk_last + off + 1 : n_keys();
    wave_kv_end = wav_last + off + 1 < n_keys() ? wav_last + off + 1 : n_keys();
  }
  if (qw >= p->Sq) wave_kv_end = 0;
  const int nt = blk_kv_end > 0 ? (blk_kv_end + kBN - 1) / kBN : 0;
  int n_full = n_keys() / kBN;                              // leading tiles that need no mask for this wave
  if (CAUSAL) {
    const int lim = qw + off + 1;                          // keys < lim are visible to EVERY row of the wave
    const int nf = lim > 0 ? lim / kBN : 0;
    n_full = nf < n_full ? nf : n_full;
  }
  if (n_full > nt) n_full = nt;

USP_TM(
  const uint64_t tm_item = __builtin_amdgcn_s_memtime();
)
  // ---- Q fragments (B operand: lane holds Q[row][16t + 8hi .. +7]), parked in the accumulator file -----------------
  u32x4 qf[2][NKT];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int row = qw + 32 * qb + l31;
    const int row_c = row < p->Sq ? row : p->Sq - 1;
    const char* qp = p->q + 2 * (b * p->q_sb + (int64_t)row_c * p->q_ss + h * p->q_sh) + 16 * hi;
#pragma unroll
    for (int t = 0; t < NKT; ++t) qf[qb][t] = *(const u32x4*)(qp + 32 * t);
  }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int t = 0; t < NKT; ++t) pin_agpr4(qf[qb][t]);

  // ---- staging: LDS-DMA with running cursors.  K tiles are issued in the order 0, 1, 2, ... and V tiles 0, 1, ..., each
  // exactly once per loop iteration (also past the last tile the item needs: rows >= Sk read as 0 through num_records,
  // a surplus tile inside the tensor is fetched and ignored), so the descriptor of the next tile is the previous one
  // advanced by two scalar additions -- no per-tile 64-bit multiplies, no clamps (the head of an iteration is otherwise
  // ~100 scalar instructions that nothing hides at one wave per SIMD).  Base and remaining bytes are 64-bit scalars: no
  // sequence length or stride is refused.
  const int64_t k_tb = (int64_t)kBN * p->k_ss * 2, v_tb = (int64_t)kBN * p->v_ss * 2;   // bytes per tile step
  const char* k_cur = p->k + 2 * (b * p->k_sb + hkv * p->k_sh);
  const char* v_cur = p->v + 2 * (b * p->v_sb + hkv * p->v_sh);
  if constexpr (SPLIT) { k_cur += 2 * (int64_t)kb * p->k_ss; v_cur += 2 * (int64_t)kb * p->v_ss; }
  int rows_kv = n_keys();                                    // valid rows from the cursors on (<= 0: lanes read 0)
  const int k_rowb = (int)p->k_ss * 2, v_rowb = (int)p->v_ss * 2;
  // (lds_w / k_step / v_step pass through an opaque asm at every use: hipcc otherwise hoists the sixteen M0 values and the
  // six scalar offsets of the pieces out of the loops as invariants and then SPILLS them -- a v_readlane plus five wait
  // states in front of every LDS-DMA; computed at the use each is one s_add / s_lshl / s_mul)
  int lds_w = wave * 4096, k_step = 4 * (int)p->k_ss * 2 - 1024, v_step = 4 * (int)p->v_ss * 2 - 1024;
  u32x4 k_rs, v_rs;                                         // descriptors of the tiles being fetched
  int k_ahead = 0;                                          // 0, or -1 once the K cursor is one tile ahead of the V cursor
  int dma_kbuf = 0, dma_vbuf = 0;
  // open the next K / V tile (descriptor for the cursor's tile, then advance the cursor); its four pieces follow
  auto dma_open = [&](int kbuf, int vbuf) {
    k_rs = make_rsrc_rows(k_cur, rows_kv + kBN * k_ahead, kBN, k_rowb, 2 * D);
    v_rs = make_rsrc_rows(v_cur, rows_kv, kBN, v_rowb, 2 * D);
    k_cur += k_tb;
    v_cur += v_tb;
    rows_kv -= kBN;
    dma_kbuf = kbuf; dma_vbuf = vbuf;
  };
  auto dma_open_k = [&](int kbuf) {                         // (the prologue's extra K tile: the K cursor then runs one tile ahead)
    k_rs = make_rsrc_rows(k_cur, rows_kv, kBN, k_rowb, 2 * D);
    k_cur += k_tb;
    k_ahead = -1;
    dma_kbuf = kbuf;
  };
  // piece n of the opened tiles: n < 4 -> K piece n, else V piece n - 4
  auto dma_piece = [&](int n) {
    asm volatile("" : "+s"(lds_w), "+s"(k_step), "+s"(v_step));
    if (n < 4) lds_dma16_asm(k_rs, lds_w + dma_kbuf * KBYTES, k_voff ^ (64 * n), n * k_step, n);
    else lds_dma16_asm(v_rs, lds_w + VOFF + dma_vbuf * KBYTES, v_voff, (n - 4) * v_step, n - 4);
  };
  auto dma_all = [&](int kbuf, int vbuf) {                  // next K tile -> Kbuf[kbuf], next V tile -> Vbuf[vbuf]
    dma_open(kbuf, vbuf);
#pragma unroll
    for (int n = 0; n < 8; ++n) dma_piece(n);
  };

  // ---- accumulators / softmax state (index = query block) -----------------------------------------------------------
  f32x16 o[2][NDJ];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int dj = 0; dj < NDJ; ++dj) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][dj][r] = 0.f;
      pin_agpr(o[qb][dj]);
    }
  float m_run[2] = {USP_NEG_INF, USP_NEG_INF};   // running row max, raw score units
  float l_run[2] = {0.f, 0.f};                   // this lane's share of the row sum

  // ---- building blocks outside the pipeline (the first tile of an item; the mask of a diagonal / ragged tile) ----------
  // S^T = K Q^T for the K tile in Kbuf[kbuf]
  auto qk = [&](int kbuf, f32x16 (&s)[2][2]) {
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      USP_LDS const char* kp = smem + kbuf * KBYTES + (k_rd ^ (32 * kt));
      const u32x4 k0 = *(USP_LDS const u32x4*)kp;
      const u32x4 k1 = *(USP_LDS const u32x4*)(kp + 32 * ROWB);
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        if (kt == 0) { M::template s_first<true>(s[qb][0], k0, qf[qb][0]); M::template s_first<true>(s[qb][1], k1, qf[qb][0]); }
        else { M::template s_next<true>(s[qb][0], k0, qf[qb][kt]); M::template s_next<true>(s[qb][1], k1, qf[qb][kt]); }
      }
    }
    mfma_settle(s);
  };
  auto mask = [&](int kt0, f32x16 (&s)[2][2]) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int row = qw + 32 * qb + l31;
      int klim = n_keys() - 1;
      if (CAUSAL) klim = row + off < klim ? row + off : klim;
      const int kb0 = kt0 + 4 * hi;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb0 + (r & 3) + 8 * (r >> 2);
        if (key > klim) s[qb][0][r] = USP_NEG_INF;
        if (key + 32 > klim) s[qb][1][r] = USP_NEG_INF;
      }
    }
  };
  // ---- prologue: K(0), V(0), K(1) resident; S(0) computed -------------------------------------------------------------
  f32x16 sa[2][2], sb[2][2];      // S^T [query block][key block] of the current / the next tile (ping-pong)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) { sa[qb][kb][r] = 0.f; sb[qb][kb][r] = 0.f; }
  dma_all(0, 0);                                 // K(0), V(0)
  dma_open_k(1);                                 // K(1)
#pragma unroll
  for (int n = 0; n < 4; ++n) dma_piece(n);
  dma_drain();
  __syncthreads();
  if (nt > 0 && wave_kv_end > 0) qk(0, sa);
  // K(0) must have been read by EVERY wave before the first loop iteration refills Kbuf[0] with K(2)
  __syncthreads();

  // ---- the tile loop: hand-pinned software pipeline, 64 MFMA slots per tile ------------------------------------------
  // Iteration jj works on TWO tiles: softmax + PV of tile jj, whose raw scores S(jj) it receives, and the scores of tile
  // jj + 1, which it hands on.
  //   phase A (32 slots): S(jj+1) = K(jj+1) Q^T, k-step major, 4 accumulators in turn (no dependent neighbours); every
  //            K fragment serves two MFMAs; beside them NEA of the 64 exp2 / row-sum / pack elements of tile jj;
  //   phase B (32 slots): O^T += V(jj)^T P(jj)^T, 8 accumulators in turn, every V fragment serves two MFMAs; the first
  //            64 - NEA slots finish tile jj's elements (k-step 3 of P is first needed by slot 24), slots from MAX0 on
  //            carry the row-max chain of S(jj+1) (v_max3), the last one the defer-max decision;
  //   the 8 LDS-DMA pieces of K(jj+2) / V(jj+1) go out one behind every DMAS-th MFMA from slot DMA0 on.
  //   Defer-max: O and l are rescaled only when some row's max grew by more than 2^kThr (wave-uniform, rare);
  //   otherwise the old reference max is kept (P <= 2^kThr).
  // sched_barrier(0) pins the slots; the asm MFMAs keep their program order among themselves.
  // MODE 0 is the steady state.  MODE 1: tile jj+1 needs the causal / ragged mask -- applied to S(jj+1) between the
  // phases, in front of its row-max chain (a wave meets one or two such tiles per item: the diagonal).  MODE 2: there is
  // no tile jj+1 for this wave -- phase A carries only the element work.  So every tile a wave needs runs through the
  // pipelined code; with the diagonal on an unpipelined path the four waves of a workgroup, which share the per-tile
  // barrier, spent the last four iterations of every item at that path's pace (causal C2: 54 % MFMA-pipe occupancy
  // against 62 % for the unmasked launch, profiles/r04a_pmc_fwd_*.txt).
  constexpr float kThr = 8.f;
  constexpr int NA = 32, NB = 32;
  constexpr int NEA = kF64_NEA, LEAD = kF64_LEAD, PFK = kF64_PFK, PFV = kF64_PFV, MAX0 = kF64_MAX0;
  constexpr int DMA0 = kF64_DMA0, DMAS = kF64_DMAS;
  static_assert(64 - NEA <= 22, "k-step 3 of P must be complete two slots before slot 24 of phase B");
  static_assert(MAX0 >= 1 && MAX0 < NB && DMAS >= 1 && DMA0 + 7 * DMAS < NA + NB, "");
  const float thr_raw = kThr / c;
  float m_thr[2] = {USP_NEG_INF, USP_NEG_INF};   // m_run + kThr / c: a tile whose scores stay below keeps the reference
  float nmc[2] = {0.f, 0.f};                     // -(reference max * c), 0 while the reference is still -inf
  // the first tile of an item: O and l are still zero, only the reference max is set (the general path below would read,
  // scale and write back all 128 accumulator registers: ~2 000 cycles per item that nothing overlaps)
  auto first_ref = [&](const float (&mt_lane)[2]) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const float m_new = xhalf_max(mt_lane[qb]);
      const float m_use = (m_new == USP_NEG_INF) ? 0.f : m_new;
      m_run[qb] = m_new;
      m_thr[qb] = m_new + thr_raw;
      nmc[qb] = -(m_use * c);
    }
  };
  auto rescale = [&](const float (&mt_lane)[2]) {
    asm volatile("; rescale (rare)" ::: "memory");          // keeps hipcc from if-converting the branch
    mfma_settle(o);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const float m_new = fmaxf(m_run[qb], xhalf_max(mt_lane[qb]));
      const float m_use = (m_new == USP_NEG_INF) ? 0.f : m_new;
      const float alpha = (m_new == m_run[qb]) ? 1.f : fast_exp2(m_run[qb] * c - m_use * c);
      m_run[qb] = m_new;
      m_thr[qb] = m_new + thr_raw;
      nmc[qb] = -(m_use * c);
      l_run[qb] *= alpha;
#pragma unroll
      for (int dj = 0; dj < NDJ; ++dj)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[qb][dj][r] *= alpha;
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int dj = 0; dj < NDJ; ++dj) pin_agpr(o[qb][dj]);
    operand_settle();                                       // v_accvgpr_write -> MFMA SrcC
  };
  // row max of a complete score tile + the defer-max decision (outside the pipeline: the first tile of an item)
  auto decide = [&](f32x16 (&s)[2][2]) {
    float mt[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      mt[qb] = s[qb][0][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mt[qb] = fmaxf(mt[qb], s[qb][0][r]);
#pragma unroll
      for (int r = 0; r < 16; ++r) mt[qb] = fmaxf(mt[qb], s[qb][1][r]);
    }
    first_ref(mt);                                           // (only ever called for the first tile of an item)
  };
  // PARV = jj & 1 as a compile-time constant (0 / 1: every LDS offset of the iteration is an immediate) or 2: taken from `jpar`
  auto iter = [&](auto par_c, auto mode_c, int jpar, int kt0_next, f32x16 (&cs)[2][2], f32x16 (&ns)[2][2])
      __attribute__((always_inline)) {
    constexpr int PARV = decltype(par_c)::value, MODE = decltype(mode_c)::value;
    const int par = PARV < 2 ? PARV : jpar;
    // K(jj+2) -> Kbuf[jj&1], which held K(jj), and V(jj+1) -> Vbuf[(jj+1)&1], which held V(jj-1): both last read in the
    // previous iteration
    dma_open(par, par ^ 1);
    auto dma_slot = [&](int g) {                              // g = slot index over both phases
      if (g >= DMA0 && (g - DMA0) % DMAS == 0 && (g - DMA0) / DMAS < 8) dma_piece((g - DMA0) / DMAS);
    };
    USP_LDS const char* kb = smem + (par ^ 1) * KBYTES;
    USP_LDS const char* vb = smem + par * KBYTES + v_rd;
    int kr = k_rd;
    asm volatile("" : "+v"(kr));     // opaque per iteration: hipcc otherwise hoists the eight k_rd ^ 32t out of the loop and spills them
    u32x4 ka[2 * NKT];                                      // fragment f = 2*kt + key block
    auto rd_k = [&](int f) {
      ka[f] = *(USP_LDS const u32x4*)(kb + (f & 1) * 32 * ROWB + (kr ^ (32 * (f >> 1))));
    };
    u32x4 va[4 * NDJ];                                      // fragment f = NDJ*ks + dj
    auto rd_v = [&](int f) {
      USP_LDS const char* vp = vb + (4 * (f / NDJ) * NDJ + (f % NDJ)) * 256;
      const u32x2 v0 = lds_read_tr16(vp);
      const u32x2 v1 = lds_read_tr16(vp + 2 * NDJ * 256);
      va[f] = u32x4{v0[0], v0[1], v1[0], v1[1]};
    };
    float rs[2] = {0.f, 0.f};
    u32x4 pf[2][4];
    // element e of the tile's 64 scores per lane, in the order the PV k-steps need them:
    //   e = 16*ks + 8*qb + r8  ->  cs[qb][ks >> 1][8*(ks & 1) + r8]
    // The consumers of an exp2 result run ONE ELEMENT LATE (row-sum add of e-1, pack of the pair (e-2, e-1)).
    auto get = [&](int e) -> float { return cs[(e >> 3) & 1][e >> 5][8 * ((e >> 4) & 1) + (e & 7)]; };
    auto consume = [&](int e) {                              // row sum (+ pack when e closes a pair)
      rs[(e >> 3) & 1] += get(e);
      if (e & 1) pf[(e >> 3) & 1][e >> 4][(e & 7) >> 1] = E::pack2(get(e - 1), get(e));
    };
    auto exp_elem = [&](int e) {
      cs[(e >> 3) & 1][e >> 5][8 * ((e >> 4) & 1) + (e & 7)] = fast_exp2(__builtin_fmaf(get(e), c, nmc[(e >> 3) & 1]));
      if (e > 0) consume(e - 1);
      if (e == 63) consume(63);
    };
    if (MODE != 2) {
#pragma unroll
      for (int f = 0; f < PFK; ++f) rd_k(f);
    }
#pragma unroll
    for (int e = 0; e < LEAD; ++e) exp_elem(e);
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- phase A ----------------
#pragma unroll
    for (int sl = 0; sl < NA; ++sl) {
      const int f = sl >> 1, kt = f >> 1, kbk = f & 1, qb = sl & 1;
      if (MODE != 2) {
        if (qb == 0 && f + PFK < 2 * NKT) rd_k(f + PFK);
        if (kt == 0) M::template s_first<MODE != 0>(ns[qb][kbk], ka[f], qf[qb][0]);
        else M::template s_next<MODE != 0>(ns[qb][kbk], ka[f], qf[qb][kt]);
      }
#pragma unroll
      for (int e = LEAD + sl * (NEA - LEAD) / NA; e < LEAD + (sl + 1) * (NEA - LEAD) / NA; ++e) exp_elem(e);
      // the V fragments of phase B's first MFMAs are read behind phase A's last ones (V(jj) has been resident since
      // the previous barrier)
      if (sl >= NA - 2 * PFV && ((sl - (NA - 2 * PFV)) & 1) == 0) rd_v((sl - (NA - 2 * PFV)) >> 1);
      dma_slot(sl);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 1) {                 // the diagonal / the ragged last tile: mask S(jj+1) before anything looks at it
      mfma_settle(ns);
      mask(kt0_next, ns);
    }
    // ---------------- phase B ----------------
    float mt[2] = {USP_NEG_INF, USP_NEG_INF};
    bool keep = true;
    constexpr int NMAX = NB - MAX0;                           // slots that carry row-max work (64 values)
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int f = i >> 1, qb = i & 1;
      if (qb == 0 && f + PFV < 4 * NDJ) rd_v(f + PFV);
      M::template o_acc<MODE != 0>(o[qb][f % NDJ], va[f], pf[qb][f / NDJ]);
      if (NEA + i < 64) exp_elem(NEA + i);
      if (i == 64 - NEA - 1 || (NEA == 64 && i == 0)) { l_run[0] += rs[0]; l_run[1] += rs[1]; }
      if (MODE != 2 && i >= MAX0) {
#pragma unroll
        for (int x = (i - MAX0) * 64 / NMAX; x < (i - MAX0 + 1) * 64 / NMAX; ++x)      // value x: qb = x >> 5
          mt[x >> 5] = fmaxf(mt[x >> 5], ns[x >> 5][(x >> 4) & 1][x & 15]);
        if (i == NB - 1) keep = __all(mt[0] <= m_thr[0] && mt[1] <= m_thr[1]);       // behind the last MFMA, not after it
      }
      dma_slot(NA + i);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!keep) rescale(mt);
    dma_drain();                 // this wave's pieces of K(jj+2), V(jj+1) have landed ...
    __syncthreads();             // ... and so have everybody else's
  };

  // this wave's tiles: [0, n_w); [0, n_full) need no mask.  Iteration jj is plain while jj + 1 < n_full.
  const int n_w = wave_kv_end > 0 ? (wave_kv_end + kBN - 1) / kBN : 0;
  if (n_full > n_w) n_full = n_w;
  int j = 0;
USP_TM(
  const uint64_t tm_loop = __builtin_amdgcn_s_memtime();
  uint64_t tm_hot = tm_loop;
)
  if (n_w > 0) {
    if (n_full == 0) mask(0, sa);
    decide(sa);
    const std::integral_constant<int, 0> c0;
    const std::integral_constant<int, 1> c1;
    const std::integral_constant<int, 2> c2;
    const int n_hot = n_full - 1;                            // iterations [0, n_hot) are plain
    for (; j + 1 < n_hot; j += 2) {
      iter(c0, c0, 0, 0, sa, sb);
      iter(c1, c0, 1, 0, sb, sa);
    }
    auto adopt = [&]() {                                     // the scores handed on become the current ones
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) sa[qb][kb] = sb[qb][kb];
    };
USP_TM(
    tm_hot = __builtin_amdgcn_s_memtime();
)
    if (j < n_hot) { iter(c0, c0, 0, 0, sa, sb); adopt(); ++j; }
    for (; j < n_w; ++j) {                                   // the diagonal (MODE 1) and this wave's last tile (MODE 2)
      if (j + 1 >= n_w) iter(c2, c2, j & 1, 0, sa, sb);
      else { iter(c2, c1, j & 1, (j + 1) * kBN, sa, sb); adopt(); }
    }
  }
USP_TM(
  const uint64_t tm_own = __builtin_amdgcn_s_memtime();
)
  // ---- the tiles other waves of the workgroup still work on: keep the K/V stream and the barrier cadence -------------
  for (; j < nt; ++j) {
    dma_all(j & 1, (j + 1) & 1);
    dma_drain();
    __syncthreads();
  }

USP_TM(
  const uint64_t tm_idle = __builtin_amdgcn_s_memtime();
)
  // ---- epilogue: normalise, merge with the running result, store (per query block, as in usp_flash_fwd.hip) -------
  mfma_settle(o);
  asm volatile("" : "+s"(p));
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int row = qw + 32 * qb + l31;
    const float l_tot = xhalf_sum(l_run[qb]);
    const bool empty = !(l_tot > 0.f);
    const float inv = empty ? 0.f : 1.f / l_tot;
    const float blk_lse = empty ? USP_NEG_INF : (m_run[qb] * c + log2f(l_tot)) * kLn2;
    float w_blk = inv, w_old = 0.f, new_lse = blk_lse;
    float* lse_p = p->lse + b * p->lse_sb + h * p->lse_sh + row;
    // (split form) launch_fwd64 has pointed acc / lse at cut 0 of the workspace ([ksplit][B,Sq,Hq,D] fp32 rows and
    // [ksplit][B,Hq,Sq] LSEs), never final, never merged here -- split_merge_kernel does both
    if constexpr (SPLIT) lse_p += (int64_t)ks * p->B * p->Hq * p->Sq;
    const bool valid = row < p->Sq;
    const bool fin = row >= p->final_begin && row < p->final_end;
    // single-pass call whose 32 rows are all final and 16-byte aligned: straight-line widened stores
    const bool wide = !p->merge_in && p->out_wide && __all(!valid || fin);
    if (valid) {
      if (p->merge_in) {
        const float old = *lse_p;
        const float mx = fmaxf(old, blk_lse);
        if (mx == USP_NEG_INF) {
          new_lse = USP_NEG_INF; w_old = 0.f; w_blk = 0.f;
        } else {
          const float e_old = exp2f((old - mx) * kLog2e);
          const float e_blk = exp2f((blk_lse - mx) * kLog2e);
          const float sum = e_old + e_blk;
          new_lse = mx + log2f(sum) * kLn2;
          w_old = e_old / sum;
          w_blk = e_blk / sum * inv;
        }
      }
      if (hi == 0) *lse_p = new_lse;
      int64_t arow = b * p->a_sb + (int64_t)row * p->a_ss + h * p->a_sh;
      if constexpr (SPLIT) arow += (int64_t)ks * p->B * p->Sq * p->Hq * D;
      const int64_t orow = b * p->o_sb + (int64_t)row * p->o_ss + h * p->o_sh;
      if (wide) {
        // each row is split across the two half-waves in 8-byte pieces; one v_permlane32_swap per dword regroups two
        // adjacent pieces into 16 contiguous bytes per lane (the store tail is issue-bound)
        char* op = p->out + 2 * orow;
#pragma unroll
        for (int dj = 0; dj < NDJ; ++dj)
#pragma unroll
          for (int g2 = 0; g2 < 2; ++g2) {
            const int r0 = 8 * g2;
            uint32_t ax = E::pack2(o[qb][dj][r0] * w_blk, o[qb][dj][r0 + 1] * w_blk);
            uint32_t ay = E::pack2(o[qb][dj][r0 + 2] * w_blk, o[qb][dj][r0 + 3] * w_blk);
            uint32_t bx = E::pack2(o[qb][dj][r0 + 4] * w_blk, o[qb][dj][r0 + 5] * w_blk);
            uint32_t by = E::pack2(o[qb][dj][r0 + 6] * w_blk, o[qb][dj][r0 + 7] * w_blk);
            const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
            *(u32x4*)(op + 2 * (32 * dj + 16 * g2 + 8 * hi)) = u32x4{sx[0], sy[0], sx[1], sy[1]};
          }
      } else {
#pragma unroll
        for (int dj = 0; dj < NDJ; ++dj) {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int d0 = 32 * dj + 8 * g4 + 4 * hi;
            f32x4 val = {o[qb][dj][4 * g4] * w_blk, o[qb][dj][4 * g4 + 1] * w_blk, o[qb][dj][4 * g4 + 2] * w_blk,
                         o[qb][dj][4 * g4 + 3] * w_blk};
            if (p->merge_in) {
              const f32x4 a = *(const f32x4*)(p->acc + arow + d0);
              val += a * w_old;
            }
            if (fin) {
              u32x2 pk = {E::pack2(val[0], val[1]), E::pack2(val[2], val[3])};
              *(u32x2*)(p->out + 2 * (orow + d0)) = pk;
            } else {
              *(f32x4*)(p->acc + arow + d0) = val;
            }
          }
        }
      }
    }
  }
USP_TM(
  {
    const uint64_t tm_end = __builtin_amdgcn_s_memtime();
    if (lane == 0 && (blockIdx.x % 61) == 0 && pass < 4)
      printf("TF wg %3d pass %d wave %d qt %2d tiles own %3d wg %3d : prologue %6llu hot-loop %8llu (%5llu / tile) tail-tiles %6llu idle %6llu epilogue %6llu\n",
             (int)blockIdx.x, pass, wave, qt, n_w, nt, (unsigned long long)(tm_loop - tm_item), (unsigned long long)(tm_hot - tm_loop),
             (unsigned long long)((tm_hot - tm_loop) / (n_full > 2 ? (n_full - 1) / 2 * 2 : 1)), (unsigned long long)(tm_own - tm_hot),
             (unsigned long long)(tm_idle - tm_own), (unsigned long long)(tm_end - tm_idle));
  }
)
  }  // next item
}

template <bool SPLIT>
static void launch64(const FwdArgsT<SPLIT>& p, int grid, size_t lds, int dtype, bool causal, hipStream_t st) {
  if (dtype == USP_BF16) {
    if (causal) hipLaunchKernelGGL((flash_fwd64_kernel<0, true, SPLIT>), dim3(grid), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((flash_fwd64_kernel<0, false, SPLIT>), dim3(grid), dim3(256), lds, st, p);
  } else {
    if (causal) hipLaunchKernelGGL((flash_fwd64_kernel<1, true, SPLIT>), dim3(grid), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((flash_fwd64_kernel<1, false, SPLIT>), dim3(grid), dim3(256), lds, st, p);
  }
}

bool launch_fwd64(const FwdArgsT<true>& p_in, int dtype, bool causal, hipStream_t st, int* rc) {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  // the K pieces' swizzle is XORed into the per-lane byte offset (row part a multiple of 256 bytes); per-lane offsets and the
  // pieces' scalar offsets are 32-bit: 64 rows of K / V must span less than 2^31 bytes
  if ((p_in.k_ss * 2) % 256 != 0 || p_in.k_ss * 128 >= (1LL << 31) || p_in.v_ss * 128 >= (1LL << 31)) return false;
  if (p_in.win_on || p_in.seq_q) return false;
  const size_t lds = 2 * 2 * kBN * 128 * 2;
  if (p_in.ksplit > 1) {
    FwdArgsT<true> p = p_in;
    p.nq = (p.Sq + 255) / 256;
    p.n_items = p.B * p.Hq * p.nq * p.ksplit;
    const int grid = (!p.interleave && p.n_items > cus) ? cus : p.n_items;
    // the cuts write partials: the epilogue's fp32 destination is cut 0 of the workspace (the kernel adds the cut)
    p.acc = p.ws_o;
    p.a_sb = (int64_t)p.Sq * p.Hq * 128; p.a_ss = (int64_t)p.Hq * 128; p.a_sh = 128;
    p.lse = p.ws_lse;
    p.lse_sb = (int64_t)p.Hq * p.Sq; p.lse_sh = p.Sq;
    p.merge_in = 0; p.final_begin = 0; p.final_end = 0; p.out_wide = 0;
    launch64<true>(p, grid, lds, dtype, causal, st);
    *rc = hipGetLastError() == hipSuccess ? launch_split_merge(p_in, dtype, 128, st) : USP_ELAUNCH;
    return true;
  }
  FwdArgsT<false> p;
  static_cast<FwdParams&>(p) = p_in;
  p.nq = (p.Sq + 255) / 256;
  p.n_items = p.B * p.Hq * p.nq;
  const int grid = (!p.interleave && p.n_items > cus) ? cus : p.n_items;      // persistent: one workgroup per CU
  launch64<false>(p, grid, lds, dtype, causal, st);
  *rc = hipGetLastError() == hipSuccess ? USP_OK : USP_ELAUNCH;
  return true;
}

}  // namespace usp
