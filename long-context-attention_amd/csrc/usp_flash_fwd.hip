// Blockwise flash-attention forward for gfx950 with the ring LSE-merge fused into the epilogue.
// C ABI: usp_flash_fwd (include/usp_hip.h).  Replaces the reference's `fwd-only` block kernel
// (yunchang/kernels/attention.py:44-136) and update_out_and_lse (yunchang/ring/utils.py:10-51).
//
// Structure (one workgroup = 8 waves = 256 query rows of one (batch, head); KV tile = 64 keys):
//   * "fully swapped" MFMA orientation: S^T = K Q^T and O^T = V^T P^T with
//     v_mfma_f32_32x32x16, so that every lane owns ONE query row (q = lane & 31) in both
//     accumulators; the two half-waves hold complementary key / dim subsets.  Row max / sum /
//     rescale / LSE merge are therefore lane-local (+ one v_permlane32_swap per row statistic).
//   * Q fragments live in registers for the whole kernel (B operand of K Q^T).
//   * K tile in LDS row-major with a 16-byte-slot XOR swizzle -> conflict-free ds_read_b128.
//   * V tile in LDS as [4 keys][32 dims] 256-byte blocks -> ds_read_b64_tr_b16 delivers the
//     A operand of V^T P^T directly; one block = exactly one LDS bank row per half-wave.
//   * P needs no cross-lane shuffle: the key order of each PV k-step is DEFINED as the order in
//     which the S^T accumulator holds keys, and V is read in that same order.
//   * K/V tiles are double-buffered in LDS and filled by LDS-DMA (buffer_load ... lds) one tile (V) /
//     two tiles (K) ahead: no staging registers, no ds_write; one barrier per tile.
//   * exp2 with softmax_scale*log2(e) folded into one FMA per score.
#include <stdlib.h>

#include "usp_common.hpp"
#include "usp_fwd_params.hpp"
#include "usp_hip.h"

namespace usp {

// NWAVES waves per workgroup, 32 query rows each: 8 (one 256-row workgroup per CU) or 4 (two 128-row
// workgroups per CU: half the causal diagonal waste, and the two workgroups desynchronise).
// KSPLIT: the K-split variant is its own instantiation -- the plain kernels sit on the register cliff (256 VGPRs), and
// with the split code compiled in unconditionally hipcc spilled 16 more bytes in them.
template <int D, int DT, bool CAUSAL, int NWAVES, bool KSPLIT = false>
__global__ __launch_bounds__(64 * NWAVES, 2) void flash_fwd_kernel(const FwdArgsT<KSPLIT> p_in) {
  using E = Elem<DT>;
  constexpr int kThreads = 64 * NWAVES;
  constexpr int kBM = 32 * NWAVES;
  constexpr int ROWB = D * 2;                 // bytes per K row
  constexpr int KBYTES = kBN * ROWB;          // one K (or V) tile
  constexpr int NKT = D / 16;                 // k-steps of K Q^T
  constexpr int NDJ = D / 32;                 // 32-wide dim tiles of O^T
  // LDS: Kbuf[0], Kbuf[1], Vbuf[0], Vbuf[1]
  constexpr int VOFF = 2 * KBYTES;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  USP_LDS char* smem = (USP_LDS char*)smem_raw;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31;
  const int hi = lane >> 5;

  // ---- persistent workgroups: each walks a static list of (batch, head, query tile) items (ItemWalk) --
  const ItemWalk walk(p_in.n_items);
  ItemQueue queue{p_in.sched, p_in.seq_q, p_in.B * p_in.Hq, p_in.nq, p_in.Hq, kBM, CAUSAL ? 1 : 0};
  int qstate = 0;
  USP_LDS int* qslots = (USP_LDS int*)(smem + 4 * KBYTES);          // 2 ints behind the K/V buffers
  for (int pass = 0;; ++pass) {
  int w = p_in.sched ? item_queue_next(queue, qstate, qslots, pass) : walk.at(pass);
  if (w < 0) break;
  FwdParams p = p_in;
  int ks = 0;
  if constexpr (KSPLIT) {
    w = walk.dealt(w, p.nq * p_in.ksplit);                 // the K cuts of a tile are dealt like tiles
    ks = w % p_in.ksplit; w /= p_in.ksplit;
  } else {
    if (!p_in.sched) w = walk.dealt(w, p.nq);
  }
  const int qt_r = w % p.nq;
  int rest = w / p.nq;
  const int qt = CAUSAL ? (p.nq - 1 - qt_r) : qt_r;      // heavy (late) tiles first
  const int g = rest % p.G;
  rest /= p.G;
  const int hkv = rest % p.Hkv;
  const int b = rest / p.Hkv;
  const int h = hkv * p.G + g;

  // ---- packed variable-length batch: bind this workgroup to the rows of sequence b -------------------
  // (the host passes batch strides of 0 in this mode, so every `b * stride_b` below vanishes)
  if (p.seq_q != nullptr) {
    const int q_first = p.seq_q[2 * b], q_len = p.seq_q[2 * b + 1];
    const int k_first = p.seq_k[2 * b], k_len = p.seq_k[2 * b + 1];
    if (qt * kBM >= q_len) continue;                        // whole workgroup past the end of its sequence
    p.q += 2 * q_first * p.q_ss;
    p.k += 2 * k_first * p.k_ss;
    p.v += 2 * k_first * p.v_ss;
    if (p.out) p.out += 2 * q_first * p.o_ss;
    if (p.acc) p.acc += q_first * p.a_ss;
    p.lse += q_first;
    p.Sq = q_len;
    p.Sk = k_len > 0 ? k_len : 0;
    p.causal_off = p.Sk - q_len;
    const int half = q_len >> 1;                            // final_begin/_end count half sequences here
    p.final_begin = p.final_begin >= 2 ? q_len : p.final_begin * half;
    p.final_end = p.final_end >= 2 ? q_len : p.final_end * half;
  }

  // ---- K split: bind this workgroup to cut `ks` of the keys its query tile sees ------------------------
  // Few (batch, head, tile) items cannot fill the part, and a causal launch lasts as long as its heaviest item: the
  // tiles [0, nt) of the item are cut into ksplit equal runs, one workgroup each, by rebasing the K/V pointers, Sk
  // and the causal offset (the tile loops are untouched, as in packed mode).  Each cut writes its own normalised
  // partial (fp32) and its LSE to the workspace through the ordinary not-final epilogue; split_merge_kernel combines
  // them (and the running result, and the 16-bit emission) afterwards.
  bool win = false;       // left window bound active (split instantiation only)
  int win_lo = 0;         // row i sees key j only if j >= i + win_lo (in the rebased key numbering)
  if constexpr (KSPLIT) {
    win = p_in.win_on != 0;
    win_lo = p_in.win_lo;
    const int q0s = qt * kBM;
    int e = p.Sk;
    if (CAUSAL) {
      const int lim = (q0s + kBM < p.Sq ? q0s + kBM : p.Sq) + p.causal_off;
      e = lim < e ? lim : e;
    }
    const int nt_all = e > 0 ? (e + kBN - 1) / kBN : 0;
    int t0 = 0;           // first tile any row of this query tile sees (its first row has the smallest left bound)
    if (win) {
      const int first = q0s + win_lo;
      t0 = first > 0 ? first / kBN : 0;
      t0 = t0 < nt_all ? t0 : nt_all;
    }
    const int ntw = nt_all - t0;
    const int kb = (t0 + ks * ntw / p_in.ksplit) * kBN;
    int ke = (ks == p_in.ksplit - 1) ? p.Sk : (t0 + (ks + 1) * ntw / p_in.ksplit) * kBN;
    ke = ke < p.Sk ? ke : p.Sk;
    p.k += 2 * (int64_t)kb * p.k_ss;
    p.v += 2 * (int64_t)kb * p.v_ss;
    p.Sk = ke > kb ? ke - kb : 0;
    p.causal_off -= kb;
    win_lo -= kb;
    if (p_in.ksplit > 1) {
      p.acc = p_in.ws_o + (int64_t)ks * p.B * p.Sq * p.Hq * D;
      p.a_sb = (int64_t)p.Sq * p.Hq * D; p.a_ss = (int64_t)p.Hq * D; p.a_sh = D;
      p.lse = p_in.ws_lse + (int64_t)ks * p.B * p.Hq * p.Sq;
      p.lse_sb = (int64_t)p.Hq * p.Sq; p.lse_sh = p.Sq;
      p.merge_in = 0; p.final_begin = 0; p.final_end = 0; p.out_wide = 0;
    }
  }

  const int q0 = qt * kBM;
  const int qw = q0 + wave * 32;
  const int row = qw + l31;
  const int row_c = row < p.Sq ? row : p.Sq - 1;
  const int off = p.causal_off;

  // ---- KV range -----------------------------------------------------------------------------
  int blk_kv_end = p.Sk, wave_kv_end = p.Sk;
  if (CAUSAL) {
    const int blk_last = (q0 + kBM < p.Sq ? q0 + kBM : p.Sq) - 1;
    const int wav_last = (qw + 32 < p.Sq ? qw + 32 : p.Sq) - 1;
    blk_kv_end = blk_last + off + 1 < p.Sk ? blk_last + off + 1 : p.Sk;
    wave_kv_end = wav_last + off + 1 < p.Sk ? wav_last + off + 1 : p.Sk;
  }
  if (qw >= p.Sq) wave_kv_end = 0;
  const int nt = blk_kv_end > 0 ? (blk_kv_end + kBN - 1) / kBN : 0;
  // leading tiles that need neither a causal nor a ragged mask for this wave
  int n_full = p.Sk / kBN;
  if (CAUSAL) {
    const int lim = qw + off + 1;                         // keys < lim are visible to EVERY row of the wave
    const int nf = lim > 0 ? lim / kBN : 0;
    n_full = nf < n_full ? nf : n_full;
  }
  if (qw + 32 > p.Sq) n_full = 0;                         // ragged / inactive waves take the generic loop
  if constexpr (KSPLIT) { if (win) n_full = 0; }          // a left window bound: every tile through the masked loop
  if (n_full > nt) n_full = nt;

  // ---- Q fragments (B operand: lane holds Q[row][16t + 8hi .. +7]) ---------------------------
  u32x4 qf[NKT];
  {
    const char* qp = p.q + 2 * (b * p.q_sb + (int64_t)row_c * p.q_ss + h * p.q_sh) + 16 * hi;
#pragma unroll
    for (int t = 0; t < NKT; ++t) qf[t] = *(const u32x4*)(qp + 32 * t);
  }

  // ---- staging: LDS-DMA (buffer_load ... lds): no staging registers, no ds_write ---------------------
  // One wave-instruction fills 1 KiB of LDS linearly (wave-uniform base + lane*16):
  //   K tile (row-major, slot swizzle): 1024/ROWB whole rows; the lane landing on physical slot p of row
  //     r fetches logical slot p ^ swz(r) of that row (swizzle applied on the SOURCE side);
  //   V tile ([4 keys][32 dims] blocks of 256 B): 4 blocks; lane l fetches key 4*kb + (l%16)/4,
  //     dims 32*dj + 8*(l%4) .. +7 of block (kb, dj) = (4*piece + l/16) / NDJ, % NDJ.
  // The tile offset is folded into the 64-bit descriptor base (no 32-bit overflow at any sequence
  // length); num_records makes rows >= Sk read as 0 (they are masked).  hipcc drains the DMA
  // (vmcnt(0)) in front of the s_barrier that ends the iteration.
  const char* kbase = p.k + 2 * (b * p.k_sb + hkv * p.k_sh);
  const char* vbase = p.v + 2 * (b * p.v_sb + hkv * p.v_sh);
  constexpr int NW = kThreads / 64;
  constexpr int CHUNKS = KBYTES / 1024;           // 1 KiB pieces per tile
  constexpr int CPW = (CHUNKS + NW - 1) / NW;     // pieces per wave per tile
  int k_voff[CPW], v_voff[CPW];
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int cidx = wave + NW * i;
    {
      const int r = cidx * (1024 / ROWB) + lane / (D / 8);
      const int c8 = (lane % (D / 8)) ^ KSwz<D>::of(r);
      k_voff[i] = r * (int)p.k_ss * 2 + c8 * 16;
    }
    {
      const int blk = 4 * cidx + (lane >> 4);
      const int kb = blk / NDJ, dj = blk % NDJ;
      const int key = 4 * kb + ((lane & 15) >> 2), d = 32 * dj + 8 * (lane & 3);
      v_voff[i] = key * (int)p.v_ss * 2 + d * 2;
    }
  }
  auto tile_rsrc = [&](const char* base, int64_t ss, int tile) {
    const int64_t toff = (int64_t)tile * kBN * ss * 2;
    int64_t rem = ((int64_t)(p.Sk - 1 - tile * kBN) * ss + D) * 2;
    rem = rem < 0 ? 0 : (rem > 0xffffffffLL ? 0xffffffffLL : rem);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + toff), 0, (int)(uint32_t)rem, 0x00020000);
  };
  // piece i (of CPW) of K tile `tile` -> Kbuf[buf]; likewise V -> Vbuf[buf]
  auto dma_k = [&](int tile, int buf) {
    const auto rs_ = tile_rsrc(kbase, p.k_ss, tile);
#pragma unroll
    for (int i = 0; i < CPW; ++i)
      if (CHUNKS % NW == 0 || wave + NW * i < CHUNKS)
        lds_dma16(rs_, smem + buf * KBYTES + (wave + NW * i) * 1024, k_voff[i]);
  };
  auto dma_v = [&](int tile, int buf) {
    const auto rs_ = tile_rsrc(vbase, p.v_ss, tile);
#pragma unroll
    for (int i = 0; i < CPW; ++i)
      if (CHUNKS % NW == 0 || wave + NW * i < CHUNKS)
        lds_dma16(rs_, smem + VOFF + buf * KBYTES + (wave + NW * i) * 1024, v_voff[i]);
  };

  // ---- per-lane LDS read bases -----------------------------------------------------------------
  const int k_rd_row = l31 * ROWB;
  const int k_rd_x = hi ^ KSwz<D>::of(l31);          // (2t + hi) ^ s == (2t) ^ (hi ^ s)
  const int v_rd = VOFF + hi * NDJ * 256 + ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 +
                   (lane & 3) * 8;

  // ---- accumulators ----------------------------------------------------------------------------
  f32x16 o[NDJ];
#pragma unroll
  for (int dj = 0; dj < NDJ; ++dj)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dj][r] = 0.f;
  float m_run = USP_NEG_INF;   // running row max, raw score units
  float l_run = 0.f;           // this lane's share of the row sum
  const float c = p.scale_log2;

  // S^T = K Q^T for the K tile in Kbuf[kbuf]
  auto qk = [&](int kbuf, f32x16& s0, f32x16& s1) {
    USP_LDS const char* kb = smem + kbuf * KBYTES;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const int slot = ((2 * kt) ^ k_rd_x) * 16;
      u32x4 ka = *(USP_LDS const u32x4*)(kb + k_rd_row + slot);
      u32x4 kc = *(USP_LDS const u32x4*)(kb + 32 * ROWB + k_rd_row + slot);
      s0 = E::mfma(ka, qf[kt], s0);
      s1 = E::mfma(kc, qf[kt], s1);
    }
  };
  // online softmax of one 64-key tile held in (s0, s1); rescales o, returns P packed for the PV MFMAs
  auto softmax = [&](f32x16& s0, f32x16& s1, u32x4 (&pf)[4]) {
    float mt = s0[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s0[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s1[r]);
    mt = xhalf_max(mt);
    const float m_new = fmaxf(m_run, mt);
    const float m_use = (m_new == USP_NEG_INF) ? 0.f : m_new;
    const float mc = m_use * c;
    const float alpha = fast_exp2(m_run * c - mc);
    m_run = m_new;
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s0[r] = fast_exp2(__builtin_fmaf(s0[r], c, -mc));
      s1[r] = fast_exp2(__builtin_fmaf(s1[r], c, -mc));
      rs += s0[r] + s1[r];
    }
    l_run = l_run * alpha + rs;
#pragma unroll
    for (int dj = 0; dj < NDJ; ++dj)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dj][r] *= alpha;
    // P (B operand of V^T P^T): k-step ks = 2*n32 + (r>>3), element e = r & 7
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pf[0][j] = E::pack2(s0[2 * j], s0[2 * j + 1]);
      pf[1][j] = E::pack2(s0[8 + 2 * j], s0[8 + 2 * j + 1]);
      pf[2][j] = E::pack2(s1[2 * j], s1[2 * j + 1]);
      pf[3][j] = E::pack2(s1[8 + 2 * j], s1[8 + 2 * j + 1]);
    }
  };
  // O^T += V^T P^T for the V tile in Vbuf[vbuf]
  auto pv = [&](int vbuf, const u32x4 (&pf)[4]) {
    USP_LDS const char* vb = smem + vbuf * KBYTES + v_rd;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int dj = 0; dj < NDJ; ++dj) {
        USP_LDS const char* vp = vb + (4 * ks * NDJ + dj) * 256;
        u32x2 v0 = lds_read_tr16(vp);
        u32x2 v1 = lds_read_tr16(vp + 2 * NDJ * 256);
        u32x4 va = {v0[0], v0[1], v1[0], v1[1]};
        o[dj] = E::mfma(va, pf[ks], o[dj]);
      }
    }
  };
  auto mask = [&](int kt0, f32x16& s0, f32x16& s1) {
    int klim = p.Sk - 1;
    if (CAUSAL) klim = row + off < klim ? row + off : klim;
    const int kb0 = kt0 + 4 * hi;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kb0 + (r & 3) + 8 * (r >> 2);
      if (key > klim) s0[r] = USP_NEG_INF;
      if (key + 32 > klim) s1[r] = USP_NEG_INF;
    }
    if constexpr (KSPLIT) {
      if (win) {
        const int klo = row + win_lo;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb0 + (r & 3) + 8 * (r >> 2);
          if (key < klo) s0[r] = USP_NEG_INF;
          if (key + 32 < klo) s1[r] = USP_NEG_INF;
        }
      }
    }
  };

  // ---- prologue: K(0), V(0), K(1) resident; S(0) computed ------------------------------------------
  f32x16 sa, sb;          // S^T of the tile the softmax works on next
#pragma unroll
  for (int r = 0; r < 16; ++r) { sa[r] = 0.f; sb[r] = 0.f; }
  if (nt > 0) {
    dma_k(0, 0); dma_v(0, 0);
    if (nt > 1) dma_k(1, 1);
  }
  dma_drain();
  __syncthreads();
  if (nt > 0 && wave_kv_end > 0) qk(0, sa, sb);
  // K(0) must have been read by EVERY wave before the first loop iteration refills Kbuf[0] with K(2): a
  // wave that skips qk(0) (no valid rows) reaches that DMA at once, and a mostly out-of-range K(2) tile
  // (short sequences) lands immediately -- observed as rare small errors on ragged shapes.
  __syncthreads();

  // ---- main loop over unmasked tiles, hand-pinned software pipeline --------------------------------
  // Per iteration j (reference max m_run already decided for tile j):
  //   phase A: 2*NKT MFMAs of S(j+1) = K(j+1) Q^T, each followed by a slice of the exp2 / row-sum /
  //            pack work of tile j (24 of its 32 elements), K fragments prefetched two k-steps ahead;
  //   phase B: 4*NDJ MFMAs of O^T += V(j)^T P(j)^T, the first half each followed by one of the 8
  //            remaining exp2 elements, all of them by a slice of the row-max chain of S(j+1),
  //            V fragments prefetched two MFMAs ahead;
  //   decision: defer-max -- O and l are rescaled only when some row's max grew by more than 2^kThr
  //            (wave-uniform, rare); otherwise the old reference max is kept (P <= 2^kThr).
  // sched_barrier(0) pins the order: hipcc otherwise emits all MFMAs, then all VALU (measured).
  // The two S register sets ping-pong (no copies); the first MFMA of each chain takes C = 0.
  // K/V tiles are fetched with buffer loads: per-thread offsets are loop invariant, the tile offset
  // is a scalar (no 64-bit VALU address math in the loop).
  constexpr float kThr = 8.f;
  constexpr int NA = 2 * NKT, NB = 4 * NDJ;
  int j = 0;
  const int n_main = n_full < nt - 1 ? n_full : nt - 1;    // j + 1 < nt holds inside: no branches
  // m_thr = m_run + kThr / c: a tile whose scores all stay below it keeps the reference max.  Every lane tests the
  // maximum of ITS 32 scores of the row (the other half-wave's lane tests the other 32), so the common path needs no
  // exchange between the half-waves; the (rare, wave-uniform) rescale does it.
  const float thr_raw = kThr / c;
  float m_thr = USP_NEG_INF;
  float nmc = 0.f;             // -(reference max * c) of the pipelined loop, 0 while the reference is still -inf
  auto rescale = [&](float mt_lane) {
    asm volatile("; rescale (rare)" ::: "memory");          // keeps hipcc from if-converting the branch
    const float m_new = fmaxf(m_run, xhalf_max(mt_lane));
    const float m_use = (m_new == USP_NEG_INF) ? 0.f : m_new;
    const float alpha = fast_exp2(m_run * c - m_use * c);
    m_run = m_new;
    m_thr = m_new + thr_raw;
    nmc = -(m_use * c);
    l_run *= alpha;
#pragma unroll
    for (int dj = 0; dj < NDJ; ++dj)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dj][r] *= alpha;
  };
  // one pipelined iteration: softmax + PV of tile jj (scores in ca/cb), scores of tile jj+1 into na/nb
  auto iter = [&](int jj, f32x16& ca, f32x16& cb, f32x16& na, f32x16& nb) {
    const int jk = jj + 2 < nt ? jj + 2 : nt - 1;           // clamped prefetch (redundant load at the end)
    dma_k(jk, jj & 1);            // Kbuf[jj&1] held K(jj): last read in the previous iteration
    dma_v(jj + 1, (jj + 1) & 1);  // Vbuf[(jj+1)&1] held V(jj-1): last read in the previous iteration
    // ---------------- phase A ----------------
    USP_LDS const char* kb = smem + ((jj + 1) & 1) * KBYTES + k_rd_row;
    u32x4 ka[NKT], kc[NKT];
    auto rd_k = [&](int kt) {
      const int slot = ((2 * kt) ^ k_rd_x) * 16;
      ka[kt] = *(USP_LDS const u32x4*)(kb + slot);
      kc[kt] = *(USP_LDS const u32x4*)(kb + 32 * ROWB + slot);
    };
    constexpr int PF = 2;                                       // LDS fragment prefetch distance (k-steps / MFMAs)
#pragma unroll
    for (int t = 0; t < PF && t < NKT; ++t) rd_k(t);
    float rs = 0.f;
    u32x4 pf[4];
    // element e of the tile's 32 scores: e < 16 -> ca[e], else cb[e - 16]
    auto exp_elem = [&](int e) {
      // The consumers of an exp2 result run ONE ELEMENT LATE (the row-sum add of element e-1 and the pack of the pair
      // (e-2, e-1) are issued with element e): nothing waits for the transcendental it was just issued behind.
      auto P = [&](int i) -> float { return i < 16 ? ca[i] : cb[i - 16]; };
      if (e < 16) ca[e] = fast_exp2(__builtin_fmaf(ca[e], c, nmc));
      else cb[e - 16] = fast_exp2(__builtin_fmaf(cb[e - 16], c, nmc));
      if (e == 1) rs = P(0);
      else if (e > 1) rs += P(e - 1);
      if (e >= 2 && (e & 1) == 0) {                         // pair (e-2, e-1) complete -> pack
        const int r = (e - 2) & 15;
        if (e - 2 < 16) pf[r >> 3][(r & 7) >> 1] = E::pack2(ca[r], ca[r + 1]);
        else pf[2 + (r >> 3)][(r & 7) >> 1] = E::pack2(cb[r], cb[r + 1]);
      }
      if (e == 31) {
        rs += cb[15];
        pf[3][3] = E::pack2(cb[14], cb[15]);
      }
    };
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    USP_LDS const char* vb = smem + (jj & 1) * KBYTES + v_rd;
    u32x4 va[NB];
    auto rd_v = [&](int i) {                                  // i = ks * NDJ + dj
      const int ks = i / NDJ, dj = i % NDJ;
      USP_LDS const char* vp = vb + (4 * ks * NDJ + dj) * 256;
      const u32x2 v0 = lds_read_tr16(vp);
      const u32x2 v1 = lds_read_tr16(vp + 2 * NDJ * 256);
      va[i] = u32x4{v0[0], v0[1], v1[0], v1[1]};
    };
    // The first MFMA of the phase waits for the K fragments read just above (K(jj+1) is only guaranteed behind the
    // barrier): the first slice of exp work goes IN FRONT of it, every later slice behind the MFMA before it; the V
    // fragments of phase B's first MFMAs are read behind phase A's last ones (V(jj) has been resident since the
    // previous barrier), so phase B starts without an LDS round trip.
    constexpr int LEAD = 24 / NA;        // exp elements issued in front of the first MFMA of phase A
#pragma unroll
    for (int e = 0; e < LEAD; ++e) exp_elem(e);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sl = 0; sl < NA; ++sl) {
      const int kt = sl >> 1;
      if ((sl & 1) == 0) {
        if (kt + PF < NKT) rd_k(kt + PF);
        na = E::mfma(ka[kt], qf[kt], kt == 0 ? zero : na);
      } else {
        nb = E::mfma(kc[kt], qf[kt], kt == 0 ? zero : nb);
      }
      if (sl + 1 < NA) {
#pragma unroll
        for (int e = LEAD + sl * (24 - LEAD) / (NA - 1); e < LEAD + (sl + 1) * (24 - LEAD) / (NA - 1); ++e) exp_elem(e);
      }
      if (sl >= NA - PF && sl - (NA - PF) < NB) rd_v(sl - (NA - PF));
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---------------- phase B ----------------
    float mt = USP_NEG_INF;
    bool keep = true;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if (i + PF < NB) rd_v(i + PF);
      o[i % NDJ] = E::mfma(va[i], pf[i / NDJ], o[i % NDJ]);
      if (i < NB / 2) {
#pragma unroll
        for (int e = 24 + i * 16 / NB; e < 24 + (i + 1) * 16 / NB; ++e) exp_elem(e);
        if (i == NB / 2 - 1) l_run += rs;
      } else {                                                // row-max chain of S(jj+1), second half of the phase
#pragma unroll
        for (int e = (i - NB / 2) * 64 / NB; e < (i - NB / 2 + 1) * 64 / NB; ++e)
          mt = fmaxf(mt, e < 16 ? na[e] : nb[e - 16]);
        if (i == NB - 1) keep = __all(mt <= m_thr);           // decided behind the last MFMA, not after it
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!keep) rescale(mt);
    dma_drain();                 // this wave's pieces of K(jj+2), V(jj+1) have landed ...
    __syncthreads();             // ... and so have everybody else's
  };

  if (n_main > 0) {
    float mt = sa[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mt = fmaxf(mt, sa[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sb[r]);
    if (!__all(mt <= m_thr)) rescale(mt);
    f32x16 ta, tb;
    for (; j + 1 < n_main; j += 2) {
      iter(j, sa, sb, ta, tb);
      iter(j + 1, ta, tb, sa, sb);
    }
    if (j < n_main) {
      iter(j, sa, sb, ta, tb);
      sa = ta; sb = tb;
      ++j;
    }
  }
  // ---- generic tail: masked and/or inactive tiles ---------------------------------------------------
  for (; j < nt; ++j) {
    const int kt0 = j * kBN;
    if (j + 2 < nt) dma_k(j + 2, j & 1);
    if (j + 1 < nt) dma_v(j + 1, (j + 1) & 1);
    f32x16 na, nb;
#pragma unroll
    for (int r = 0; r < 16; ++r) { na[r] = 0.f; nb[r] = 0.f; }
    if (j + 1 < nt && kt0 + kBN < wave_kv_end) qk((j + 1) & 1, na, nb);
    if (kt0 < wave_kv_end) {
      bool need_mask = (kt0 + kBN > p.Sk) || (CAUSAL && kt0 + kBN - 1 > qw + off);
      if constexpr (KSPLIT) need_mask = need_mask || (win && kt0 < qw + 31 + win_lo);
      if (need_mask) mask(kt0, sa, sb);
      u32x4 pf[4];
      softmax(sa, sb, pf);
      pv(j & 1, pf);
    }
    sa = na; sb = nb;
    dma_drain();
    __syncthreads();
  }

  // ---- epilogue: normalise, merge with the running result, store -------------------------------
  const float l_tot = xhalf_sum(l_run);
  const bool empty = !(l_tot > 0.f);
  const float inv = empty ? 0.f : 1.f / l_tot;
  const float blk_lse = empty ? USP_NEG_INF : (m_run * c + log2f(l_tot)) * kLn2;
  float w_blk = inv, w_old = 0.f, new_lse = blk_lse;
  float* lse_p = p.lse + b * p.lse_sb + h * p.lse_sh + row;
  const bool valid = row < p.Sq;
  const bool fin = row >= p.final_begin && row < p.final_end;
  // single-pass call whose 32 rows are all final and 16-byte aligned: straight-line widened stores
  const bool wide = !p.merge_in && p.out_wide && __all(!valid || fin);
  if (valid) {
    if (p.merge_in) {
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
    const int64_t arow = b * p.a_sb + (int64_t)row * p.a_ss + h * p.a_sh;
    const int64_t orow = b * p.o_sb + (int64_t)row * p.o_ss + h * p.o_sh;
    if (wide) {
      // Each row is split across the two half-waves in 8-byte pieces; one v_permlane32_swap per dword
      // regroups two adjacent pieces into 16 contiguous bytes per lane: 2*NDJ dwordx4 stores instead of
      // 4*NDJ dwordx2 (the store tail is issue-bound; the fp32 path already stores 16 bytes per lane).
      char* op = p.out + 2 * orow;
#pragma unroll
      for (int dj = 0; dj < NDJ; ++dj)
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          const int r0 = 8 * g2;                 // regs r0..r0+3: column group 2*g2, r0+4..r0+7: group 2*g2+1
          uint32_t ax = E::pack2(o[dj][r0] * w_blk, o[dj][r0 + 1] * w_blk);
          uint32_t ay = E::pack2(o[dj][r0 + 2] * w_blk, o[dj][r0 + 3] * w_blk);
          uint32_t bx = E::pack2(o[dj][r0 + 4] * w_blk, o[dj][r0 + 5] * w_blk);
          uint32_t by = E::pack2(o[dj][r0 + 6] * w_blk, o[dj][r0 + 7] * w_blk);
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
          f32x4 val = {o[dj][4 * g4] * w_blk, o[dj][4 * g4 + 1] * w_blk, o[dj][4 * g4 + 2] * w_blk,
                       o[dj][4 * g4 + 3] * w_blk};
          if (p.merge_in) {
            const f32x4 a = *(const f32x4*)(p.acc + arow + d0);
            val += a * w_old;
          }
          if (fin) {
            u32x2 pk = {E::pack2(val[0], val[1]), E::pack2(val[2], val[3])};
            *(u32x2*)(p.out + 2 * (orow + d0)) = pk;
          } else {
            *(f32x4*)(p.acc + arow + d0) = val;
          }
        }
      }
    }
  }
  if (p_in.sched && p_in.interleave) break;   // one item per workgroup: leave room for other streams' kernels
  }  // next item
  if (p_in.sched && threadIdx.x == 0) item_queue_release(queue);
}

// Combines the partial results of a K-split launch: per query row, the `n` cuts' normalised partials (fp32) and
// LSEs [+ the running result when merge_in] -> what ONE launch would have left behind: lse, and the row in 16 bits
// (final rows) or fp32 (the others).  HBM-bound: one thread per 4 consecutive head-dim elements of a row.
template <int DT>
__global__ __launch_bounds__(256) void split_merge_kernel(const FwdArgsT<true> p, int D) {
  using E = Elem<DT>;
  const int d4 = D >> 2;
  const int64_t total = (int64_t)p.B * p.Sq * p.Hq * d4;
  const int64_t slot_o = (int64_t)p.B * p.Sq * p.Hq * D, slot_l = (int64_t)p.B * p.Hq * p.Sq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % d4);
    int64_t r = i / d4;
    const int h = (int)(r % p.Hq); r /= p.Hq;
    const int s = (int)(r % p.Sq);
    const int b = (int)(r / p.Sq);
    const int64_t l_idx = ((int64_t)b * p.Hq + h) * p.Sq + s;
    const int64_t o_idx = (((int64_t)b * p.Sq + s) * p.Hq + h) * D + 4 * c4;
    float* lse_p = p.lse + b * p.lse_sb + h * p.lse_sh + s;
    const int64_t arow = b * p.a_sb + (int64_t)s * p.a_ss + h * p.a_sh + 4 * c4;
    // The D/4 lanes of a row all read the running LSE here and lane c4 == 0 stores the new one below: a row's lanes sit
    // in ONE wavefront (launch_fwd_w asserts 64 % (D/4) == 0 and launches 256-thread blocks), so every load is issued
    // before the store in program order -- the value is read once, into a register.
    const float lse_old = p.merge_in ? *lse_p : USP_NEG_INF;
    float mx = lse_old;
    float l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      l[j] = j < p.ksplit ? p.ws_lse[j * slot_l + l_idx] : USP_NEG_INF;
      mx = fmaxf(mx, l[j]);
    }
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    float new_lse = USP_NEG_INF;
    if (mx != USP_NEG_INF) {
      float sum = 0.f, w_old = 0.f;
      if (p.merge_in) { w_old = exp2f((lse_old - mx) * kLog2e); sum = w_old; }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        l[j] = exp2f((l[j] - mx) * kLog2e);          // 0 for an empty cut (lse = -inf) and for j >= ksplit
        sum += l[j];
      }
      const float inv = 1.f / sum;
      new_lse = mx + log2f(sum) * kLn2;
      if (p.merge_in) o = *(const f32x4*)(p.acc + arow) * (w_old * inv);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < p.ksplit) o += *(const f32x4*)(p.ws_o + j * slot_o + o_idx) * (l[j] * inv);
    }
    if (c4 == 0) *lse_p = new_lse;
    if (s >= p.final_begin && s < p.final_end) {
      const u32x2 pk = {E::pack2(o[0], o[1]), E::pack2(o[2], o[3])};
      *(u32x2*)(p.out + 2 * (b * p.o_sb + (int64_t)s * p.o_ss + h * p.o_sh + 4 * c4)) = pk;
    } else {
      *(f32x4*)(p.acc + arow) = o;
    }
  }
}

int launch_split_merge(const FwdArgsT<true>& p, int dtype, int D, hipStream_t st) {
  // same stream as the cuts: the partials are complete when this starts
  if (64 % (D / 4) != 0) return USP_EUNSUPPORTED;           // (the D/4 lanes of a row must share a wavefront: D = 64, 128)
  const int64_t work = (int64_t)p.B * p.Sq * p.Hq * (D / 4);
  const int blocks = (int)((work + 255) / 256 < 4096 ? (work + 255) / 256 : 4096);
  if (dtype == USP_BF16) hipLaunchKernelGGL((split_merge_kernel<0>), dim3(blocks), dim3(256), 0, st, p, D);
  else hipLaunchKernelGGL((split_merge_kernel<1>), dim3(blocks), dim3(256), 0, st, p, D);
  return hipGetLastError() == hipSuccess ? USP_OK : USP_ELAUNCH;
}

template <int D, int DT, int NWAVES>
static int launch_fwd_w(FwdArgsT<true> p, bool causal, hipStream_t st) {
  p.nq = (p.Sq + 32 * NWAVES - 1) / (32 * NWAVES);
  p.n_items = p.B * p.Hq * p.nq * p.ksplit;
  // persistent launch: one workgroup per resident slot (8 waves: 1 per CU, 4 waves: 2 per CU)
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  static const bool persist = [] { const char* e = getenv("USP_FWD_PERSIST"); return !(e && e[0] == '0'); }();
  const int slots = cus * (NWAVES == 8 ? 1 : 2);
  const int grid = (((persist || p.sched) && !p.interleave) && p.n_items > slots) ? slots : p.n_items;
  const size_t lds = 2 * 2 * kBN * D * 2 + (p.sched ? 16 : 0);
  if (p.ksplit > 1 || p.win_on) {
    if (causal)
      hipLaunchKernelGGL((flash_fwd_kernel<D, DT, true, NWAVES, true>), dim3(grid), dim3(64 * NWAVES), lds, st, p);
    else
      hipLaunchKernelGGL((flash_fwd_kernel<D, DT, false, NWAVES, true>), dim3(grid), dim3(64 * NWAVES), lds, st, p);
  } else {
    FwdArgsT<false> plain;                                   // the argument block of the plain kernels: FwdParams alone
    static_cast<FwdParams&>(plain) = p;
    if (causal)
      hipLaunchKernelGGL((flash_fwd_kernel<D, DT, true, NWAVES>), dim3(grid), dim3(64 * NWAVES), lds, st, plain);
    else
      hipLaunchKernelGGL((flash_fwd_kernel<D, DT, false, NWAVES>), dim3(grid), dim3(64 * NWAVES), lds, st, plain);
  }
  if (hipGetLastError() != hipSuccess) return USP_ELAUNCH;
  if (p.ksplit > 1) {
    static_assert(64 % (D / 4) == 0 && 256 % (D / 4) == 0, "split_merge_kernel: the D/4 lanes of a row must share a wavefront");
    return launch_split_merge(p, DT, D, st);
  }
  return USP_OK;
}

template <int D, int DT>
static int launch_fwd(const FwdArgsT<true>& p, bool causal, hipStream_t st, int force) {
  // Workgroup shape: 8 waves (256 query rows, one workgroup per CU) stage K/V once per 256 rows and win by
  // 3-4 % whenever they can give every CU work; 4 waves (128 rows, two workgroups per CU) are used only
  // when the 8-wave item list is shorter than the CU count, or for short causal sequences (<= 1024 rows:
  // +3...8 %) (measured with persistent workgroups, profiles/).  Per call, `force` (USP_FORCE_ROW64 / USP_FORCE_WAVE32,
  // include/usp_hip.h) picks the family; per process, USP_FWD_WAVES=4|8|64 forces a shape for A/B runs (64 = the
  // 4 x 64-row kernel of usp_flash_fwd64.hip, where it applies).
  static const int forced_env = [] { const char* e = getenv("USP_FWD_WAVES"); return e ? atoi(e) : 0; }();
  const bool fwd64_ok = D == 128 && !p.seq_q && !p.win_on;   // what usp_flash_fwd64.hip serves (plain and K-split launches)
  const int split_kind = p.ksplit > 1 ? USP_KIND_FWD_SPLIT_MERGE : 0;
  if (force & USP_FORCE_ROW64) {
    int rc = USP_ELAUNCH;
    if (fwd64_ok && launch_fwd64(p, DT, causal, st, &rc)) {
      if (rc == USP_OK) launch_kinds_note(USP_KIND_FWD_ROW64 | split_kind);
      return rc;
    }
    return USP_EUNSUPPORTED;
  }
  int waves = (force & USP_FORCE_WAVE32) ? 0 : forced_env;
  if (waves != 4 && waves != 8 && waves != 64) {
    const int64_t grid8 = (int64_t)p.B * p.Hq * ((p.Sq + 255) / 256) * p.ksplit;
    // short causal sequences: less diagonal waste (dense only: in packed mode twice the items cost more to fetch)
    // beside a transfer (interleave) RCCL's resident workgroups take a few CUs: with ONE 256-row item per CU a lost
    // CU costs a whole extra round, so the 8-wave kernel halves its granularity there (kbench overlap, 8 resident copy
    // workgroups: 256 items 0.695 vs 0.718 ms; from two items per CU on the 256-row shape wins again, profiles/
    // r02_rank_emulation.txt).  That rule predates the 4 x 64 kernel, which beats the 128-row shape at one item per CU too,
    // alone and beside the copies (round 5, same harness, four launches + 3 x 16 MiB on 8 workgroups: 8192 x 16384 rows x
    // keys, 8 heads = 256 items 1.985 vs 2.081 ms; 8192 x 49152: 5.413 vs 5.773 ms; profiles/r05_fwd_small_interleave.txt):
    // where the 4 x 64 kernel serves the launch the rule no longer applies.
    waves = (grid8 < 256 || (p.interleave && grid8 < 512 && !fwd64_ok) || (!p.seq_q && causal && p.Sq <= 1024)) ? 4 : 8;
    // where the 256-row item wins, the one-wave-per-SIMD kernel (4 waves x 64 rows, usp_flash_fwd64.hip) serves it.  Cut
    // launches: a 64-row item costs more to open and close (64 Q fragments parked, 128 accumulators written as partials),
    // which pays from ~96 tiles per cut on (kbench ksplit, 8-wave split -> 4 x 64 split, merge launch included, round 5:
    // S = 65536 1 head n = 2 1174 -> 1216 TFLOP/s, 32768 2 heads n = 2 1136 -> 1162 and n = 3 1051 -> 1064, 16384 4 heads
    // n = 2 1049 -> 1057; below: 16384 4 heads n = 3 963 -> 945, n = 4 984 -> 957, 8192 6 heads n = 2 799 -> 780;
    // profiles/r05_fwd64_ksplit.txt)
    const bool long_cuts = p.ksplit <= 1 || (int64_t)p.Sk >= (int64_t)96 * kBN * p.ksplit;
    if (waves == 8 && fwd64_ok && long_cuts && !(force & USP_FORCE_WAVE32)) waves = 64;
  }
  if (waves == 64) {
    int rc = USP_ELAUNCH;
    if (fwd64_ok && launch_fwd64(p, DT, causal, st, &rc)) {
      if (rc == USP_OK) launch_kinds_note(USP_KIND_FWD_ROW64 | split_kind);
      return rc;
    }
    waves = 8;
  }
  const int rc = waves == 4 ? launch_fwd_w<D, DT, 4>(p, causal, st) : launch_fwd_w<D, DT, 8>(p, causal, st);
  if (rc == USP_OK) launch_kinds_note((waves == 4 ? USP_KIND_FWD_WAVE4 : USP_KIND_FWD_WAVE8) | split_kind);
  return rc;
}

static bool aligned16(const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; }
static bool tensor16_ok(const usp_tensor& t, int esize) {
  const int m = 16 / esize;
  return t.ptr && aligned16(t.ptr) && t.stride_b % m == 0 && t.stride_s % m == 0 &&
         t.stride_h % m == 0;
}

}  // namespace usp

extern "C" int64_t usp_flash_fwd_workspace_bytes(const usp_fwd_args* a, int32_t k_splits) {
  if (!a || k_splits <= 1 || k_splits > 8 || a->seq_q || a->seq_k) return 0;
  const int64_t rows = (int64_t)a->B * a->Sq * a->Hq;
  return (int64_t)k_splits * (rows * a->D + rows) * 4;         // partial outputs + partial LSEs, fp32 (a->D % 4 == 0)
}

extern "C" int usp_flash_fwd(const usp_fwd_args* a, void* stream) {
  using namespace usp;
  launch_kinds_reset();
  if (!a || !a->lse) return USP_EINVAL;
  const int force = a->flags & (USP_FORCE_ROW64 | USP_FORCE_WAVE32);
  if (force == (USP_FORCE_ROW64 | USP_FORCE_WAVE32)) return USP_EINVAL;
  if (a->dtype != USP_BF16 && a->dtype != USP_FP16) return USP_EINVAL;
  if (a->B <= 0 || a->Sq <= 0 || a->Sk <= 0 || a->Hq <= 0 || a->Hkv <= 0) return USP_EINVAL;
  if (!(a->softmax_scale > 0.f)) return USP_EINVAL;
  if (a->D != 32 && a->D != 64 && a->D != 128) return USP_EUNSUPPORTED;
  if (a->Hq % a->Hkv != 0) return USP_EUNSUPPORTED;
  if (!tensor16_ok(a->q, 2) || !tensor16_ok(a->k, 2) || !tensor16_ok(a->v, 2))
    return USP_EUNSUPPORTED;
  const bool packed = a->seq_q != nullptr || a->seq_k != nullptr;
  if (packed && !(a->seq_q && a->seq_k)) return USP_EINVAL;
  // sliding window (flash-attn's window_size): causal caps the right bound at 0; a right bound is the causal limit with
  // a shifted offset; a left bound runs in the split instantiation (FwdSplit)
  const bool has_win = (a->flags & USP_ATTN_WINDOW) != 0;
  const int wl = has_win ? a->window_left : -1;
  const int wr = a->causal ? 0 : (has_win ? a->window_right : -1);
  if (packed && (wl >= 0 || wr > 0)) return USP_EUNSUPPORTED;       // dense launches only
  const int f_all = packed ? 2 : a->Sq;          // packed: final_begin/_end count half sequences (0,1,2)
  int fb = a->final_begin < 0 ? 0 : a->final_begin;
  int fe = a->final_end > f_all ? f_all : a->final_end;
  if (fe < fb) fe = fb;
  const bool any_final = fe > fb, any_acc = (fb > 0 || fe < f_all);
  if (any_final && !(a->out.ptr && (reinterpret_cast<uintptr_t>(a->out.ptr) & 7) == 0 &&
                     a->out.stride_b % 4 == 0 && a->out.stride_s % 4 == 0 &&
                     a->out.stride_h % 4 == 0))
    return a->out.ptr ? USP_EUNSUPPORTED : USP_EINVAL;
  if ((any_acc || a->merge_in) && !tensor16_ok(a->acc, 4))
    return a->acc.ptr ? USP_EUNSUPPORTED : USP_EINVAL;

  FwdArgsT<true> p;
  p.q = (const char*)a->q.ptr; p.k = (const char*)a->k.ptr; p.v = (const char*)a->v.ptr;
  p.out = (char*)a->out.ptr; p.acc = (float*)a->acc.ptr; p.lse = a->lse;
  p.q_sb = a->q.stride_b; p.q_ss = a->q.stride_s; p.q_sh = a->q.stride_h;
  p.k_sb = a->k.stride_b; p.k_ss = a->k.stride_s; p.k_sh = a->k.stride_h;
  p.v_sb = a->v.stride_b; p.v_ss = a->v.stride_s; p.v_sh = a->v.stride_h;
  p.o_sb = a->out.stride_b; p.o_ss = a->out.stride_s; p.o_sh = a->out.stride_h;
  p.a_sb = a->acc.stride_b; p.a_ss = a->acc.stride_s; p.a_sh = a->acc.stride_h;
  p.lse_sb = a->lse_stride_b; p.lse_sh = a->lse_stride_h;
  p.B = a->B; p.Sq = a->Sq; p.Sk = a->Sk; p.Hq = a->Hq; p.Hkv = a->Hkv;
  p.G = a->Hq / a->Hkv;
  p.nq = 0;   // set per workgroup shape in launch_fwd_w
  p.causal_off = a->Sk - a->Sq + (wr > 0 ? wr : 0);
  p.scale = a->softmax_scale;
  p.scale_log2 = a->softmax_scale * kLog2e;
  p.merge_in = a->merge_in ? 1 : 0;
  p.final_begin = fb; p.final_end = fe;
  p.out_wide = (a->out.ptr && (reinterpret_cast<uintptr_t>(a->out.ptr) & 15) == 0 && a->out.stride_b % 8 == 0 &&
                a->out.stride_s % 8 == 0 && a->out.stride_h % 8 == 0) ? 1 : 0;
  p.seq_q = a->seq_q; p.seq_k = a->seq_k;
  p.sched = packed ? a->sched : nullptr;
  p.interleave = (a->flags & USP_LAUNCH_INTERLEAVE) ? 1 : 0;
  {   // USP_ITEM_GROUP=0 (read once): the head-major item walk of rounds 1-5 instead of a KV group's heads side by side
    static const bool group_heads = [] { const char* e = getenv("USP_ITEM_GROUP"); return !(e && e[0] == '0'); }();
    p.walk_g = group_heads ? p.G : 1;
  }
  p.ksplit = 1; p.ws_o = nullptr; p.ws_lse = nullptr;
  p.win_on = wl >= 0 ? 1 : 0; p.win_lo = a->Sk - a->Sq - (wl >= 0 ? wl : 0);
  if (a->k_splits > 1 && a->workspace != nullptr) {
    if (packed) return USP_EUNSUPPORTED;                       // dense launches only
    if (a->k_splits > 8 || !aligned16(a->workspace)) return USP_EINVAL;
    if ((any_acc || a->merge_in) && (a->acc.stride_h % 4 != 0)) return USP_EUNSUPPORTED;
    p.ksplit = a->k_splits;
    p.ws_o = (float*)a->workspace;
    p.ws_lse = p.ws_o + (int64_t)p.ksplit * a->B * a->Sq * a->Hq * a->D;
  }
  if (packed) p.q_sb = p.k_sb = p.v_sb = p.o_sb = p.a_sb = p.lse_sb = 0;
  hipStream_t st = (hipStream_t)stream;
  const bool causal = wr >= 0;                    // (a->causal, or a right window bound)
  switch (a->D * 2 + a->dtype) {
    case 32 * 2 + 0: return launch_fwd<32, 0>(p, causal, st, force);
    case 32 * 2 + 1: return launch_fwd<32, 1>(p, causal, st, force);
    case 64 * 2 + 0: return launch_fwd<64, 0>(p, causal, st, force);
    case 64 * 2 + 1: return launch_fwd<64, 1>(p, causal, st, force);
    case 128 * 2 + 0: return launch_fwd<128, 0>(p, causal, st, force);
    case 128 * 2 + 1: return launch_fwd<128, 1>(p, causal, st, force);
  }
  return USP_EUNSUPPORTED;
}
