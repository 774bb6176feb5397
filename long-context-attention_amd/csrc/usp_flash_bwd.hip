// Blockwise flash-attention backward for gfx950.  C ABI: usp_flash_bwd (include/usp_hip.h).
// Replaces the reference's `bwd-only` block kernel (yunchang/kernels/attention.py:205-250) plus the
// fp32 accumulation the ring schedules do on its results (zigzag_ring_flash_attn.py:147-170).
//
// Two launches, no atomics, deterministic:
//   MODE 0 (dQ)    : workgroup = 8 waves x 32 query rows; streams K,V tiles (64 keys) through LDS.
//                    lane owns a query row:  S^T = K Q^T, dP^T = V dO^T, dQ^T += K^T dS^T
//   MODE 1 (dK,dV) : workgroup = 4 waves x 32 keys (one wave per SIMD, 512-register budget);
//                    streams Q,dO tiles (64 rows) of every query head of the GQA group through LDS.
//                    lane owns a key:        S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS
// Both modes are one engine: two LDS tiles X1,X2 (row-major, 16-byte-slot XOR swizzle chosen so that
// BOTH ds_read_b128 row reads and ds_read_b64_tr_b16 column reads are bank-conflict free), two
// register-resident fragment sets R1,R2, S = X1 R1^T, T = X2 R2^T, and tr-read "X^T" operands for
// the gradient MFMAs.  As in the forward, no cross-lane shuffle is needed for P / dS: the k-step
// order of the gradient MFMAs is defined as the order the S accumulator holds rows.
#include "usp_common.hpp"
#include "usp_hip.h"

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
  int causal_off;
  float scale, scale_log2;
  int accum_dq, accum_dk, accum_dv;
};

constexpr int kTile = 64;           // streamed rows per LDS tile

// Swizzle of the 16-byte slot index inside a row-major [rows][D] 16-bit tile.
template <int D> USP_DEV int tile_swz(int row) {
  if (D == 128) return ((row & 3) << 2) | ((row >> 2) & 3);
  if (D == 64) return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
  return (row >> 2) & 3;   // D == 32
}

template <int D, int DT, bool CAUSAL, int MODE>
__global__ __launch_bounds__(MODE == 0 ? 512 : 256, MODE == 0 ? 2 : 1) void flash_bwd_kernel(
    const BwdParams p) {
  using E = Elem<DT>;
  constexpr int NT = MODE == 0 ? 512 : 256;     // threads
  constexpr int OWN = (NT / 64) * 32;           // rows owned by the workgroup (256 q rows / 128 keys)
  constexpr int ROWB = D * 2;
  constexpr int TILEB = kTile * ROWB;           // one streamed matrix tile
  constexpr int STATB = MODE == 1 ? 2 * kTile * 4 : 0;   // lse2 + delta of the tile's rows
  constexpr int BUFB = 2 * TILEB + STATB;
  constexpr int NKT = D / 16;
  constexpr int NDJ = D / 32;
  constexpr int NCH = kTile * D / 8;
  constexpr int NP = (NCH + NT - 1) / NT;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  USP_LDS char* smem = (USP_LDS char*)smem_raw;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31;
  const int hi = lane >> 5;
  const int off = p.causal_off;

  // ---- work item --------------------------------------------------------------------------------
  int w = xcd_remap(blockIdx.x, gridDim.x);
  const int blk_r = w % p.nblk;
  int rest = w / p.nblk;
  int b, hkv, h0, blk;
  if (MODE == 0) {
    blk = CAUSAL ? (p.nblk - 1 - blk_r) : blk_r;          // late query blocks see most keys
    const int g = rest % p.G; rest /= p.G;
    hkv = rest % p.Hkv; b = rest / p.Hkv;
    h0 = hkv * p.G + g;
  } else {
    blk = blk_r;                                           // early key blocks are seen by most rows
    hkv = rest % p.Hkv; b = rest / p.Hkv;
    h0 = hkv * p.G;
  }
  const int own0 = blk * OWN;                  // first owned row (query row / key)
  const int ow = own0 + wave * 32;             // first row owned by this wave
  const int orow = ow + l31;                   // this lane's row
  const int own_len = MODE == 0 ? p.Sq : p.Sk;
  const int orow_c = orow < own_len ? orow : own_len - 1;

  // ---- register-resident fragments R1, R2 (B operands: lane holds row[16t + 8hi .. +7]) -----------
  u32x4 r1[NKT], r2[NKT];
  {
    const char *p1, *p2;
    if (MODE == 0) {
      p1 = p.q + 2 * (b * p.q_sb + (int64_t)orow_c * p.q_ss + h0 * p.q_sh);
      p2 = p.dout + 2 * (b * p.do_sb + (int64_t)orow_c * p.do_ss + h0 * p.do_sh);
    } else {
      p1 = p.k + 2 * (b * p.k_sb + (int64_t)orow_c * p.k_ss + hkv * p.k_sh);
      p2 = p.v + 2 * (b * p.v_sb + (int64_t)orow_c * p.v_ss + hkv * p.v_sh);
    }
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      r1[t] = *(const u32x4*)(p1 + 32 * t + 16 * hi);
      r2[t] = *(const u32x4*)(p2 + 32 * t + 16 * hi);
    }
  }
  // MODE 0: lane-local row statistics
  float lse2_l = 0.f, delta_l = 0.f;
  if (MODE == 0) {
    const float l_ = p.lse[b * p.lse_sb + h0 * p.lse_sh + orow_c];
    lse2_l = (l_ == USP_NEG_INF) ? __builtin_inff() : l_ * kLog2e;
    delta_l = p.delta[b * p.dl_sb + h0 * p.dl_sh + orow_c];
  }

  // ---- streamed range ---------------------------------------------------------------------------
  // MODE 0 streams key tiles [0, nt); MODE 1 streams (head-in-group, query tile) pairs.
  const int str_len = MODE == 0 ? p.Sk : p.Sq;
  int t_begin = 0, t_end = (str_len + kTile - 1) / kTile;     // tiles per head
  if (CAUSAL) {
    if (MODE == 0) {
      const int last = (own0 + OWN < p.Sq ? own0 + OWN : p.Sq) - 1;
      const int kv_end = last + off + 1 < p.Sk ? last + off + 1 : p.Sk;
      t_end = kv_end > 0 ? (kv_end + kTile - 1) / kTile : 0;
    } else {
      const int first_q = own0 - off > 0 ? own0 - off : 0;     // first row that sees key own0
      t_begin = first_q / kTile;
      if (t_begin > t_end) t_begin = t_end;
    }
  }
  const int per_head = t_end - t_begin;
  const int n_iter = MODE == 0 ? per_head : per_head * p.G;

  // ---- staging maps -----------------------------------------------------------------------------
  int st_row[NP], st_goff[NP], st_loff[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int c = i * NT + tid;
    const int r = c / (D / 8), c8 = c % (D / 8);
    st_row[i] = r;
    st_goff[i] = c8 * 16;
    st_loff[i] = r * ROWB + ((c8 ^ tile_swz<D>(r)) * 16);
  }
  u32x4 st1[NP], st2[NP];
  float st_lse = 0.f, st_delta = 0.f;
  auto stage_load = [&](int it) {
    int tile, hh;
    if (MODE == 0) { tile = t_begin + it; hh = 0; }
    else { hh = it / per_head; tile = t_begin + it % per_head; }
    const int s0 = tile * kTile;
    const char *b1, *b2;
    int64_t ss1, ss2;
    if (MODE == 0) {
      b1 = p.k + 2 * (b * p.k_sb + hkv * p.k_sh); ss1 = p.k_ss;
      b2 = p.v + 2 * (b * p.v_sb + hkv * p.v_sh); ss2 = p.v_ss;
    } else {
      b1 = p.q + 2 * (b * p.q_sb + (h0 + hh) * p.q_sh); ss1 = p.q_ss;
      b2 = p.dout + 2 * (b * p.do_sb + (h0 + hh) * p.do_sh); ss2 = p.do_ss;
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (NCH % NT == 0 || i * NT + tid < NCH) {
        int r = s0 + st_row[i];
        r = r < str_len ? r : str_len - 1;
        st1[i] = *(const u32x4*)(b1 + 2 * (int64_t)r * ss1 + st_goff[i]);
        st2[i] = *(const u32x4*)(b2 + 2 * (int64_t)r * ss2 + st_goff[i]);
      }
    }
    if (MODE == 1 && tid < kTile) {
      const int r = s0 + tid;
      if (r < p.Sq) {
        const float l_ = p.lse[b * p.lse_sb + (h0 + hh) * p.lse_sh + r];
        st_lse = (l_ == USP_NEG_INF) ? __builtin_inff() : l_ * kLog2e;
        st_delta = p.delta[b * p.dl_sb + (h0 + hh) * p.dl_sh + r];
      } else {
        st_lse = __builtin_inff();    // rows past the end contribute P = 0
        st_delta = 0.f;
      }
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (NCH % NT == 0 || i * NT + tid < NCH) {
        *(USP_LDS u32x4*)(smem + buf * BUFB + st_loff[i]) = st1[i];
        *(USP_LDS u32x4*)(smem + buf * BUFB + TILEB + st_loff[i]) = st2[i];
      }
    }
    if (MODE == 1 && tid < kTile) {
      *(USP_LDS float*)(smem + buf * BUFB + 2 * TILEB + 4 * tid) = st_lse;
      *(USP_LDS float*)(smem + buf * BUFB + 2 * TILEB + 4 * kTile + 4 * tid) = st_delta;
    }
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
  f32x16 acc1[NDJ];                      // dQ^T (MODE 0) / dK^T (MODE 1)
  f32x16 acc2[MODE == 1 ? NDJ : 1];      // dV^T (MODE 1)
#pragma unroll
  for (int dj = 0; dj < NDJ; ++dj)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[dj][r] = 0.f; if (MODE == 1) acc2[dj][r] = 0.f; }
  const float c = p.scale_log2;

  if (n_iter > 0) { stage_load(0); stage_store(0); }
  __syncthreads();

  for (int it = 0; it < n_iter; ++it) {
    const int buf = it & 1;
    const int tile = t_begin + (MODE == 0 ? it : it % per_head);
    const int s0 = tile * kTile;                       // first streamed row of this tile
    if (it + 1 < n_iter) stage_load(it + 1);

    bool active = true, need_mask = false;
    if (MODE == 0) {
      // streamed = keys, owned = query rows
      int wave_kv_end = p.Sk;
      if (CAUSAL) {
        const int wl = (ow + 32 < p.Sq ? ow + 32 : p.Sq) - 1;
        wave_kv_end = wl + off + 1 < p.Sk ? wl + off + 1 : p.Sk;
      }
      active = ow < p.Sq && s0 < wave_kv_end;
      need_mask = (s0 + kTile > p.Sk) || (CAUSAL && s0 + kTile - 1 > ow + off);
    } else {
      // streamed = query rows, owned = keys
      active = ow < p.Sk && (!CAUSAL || (s0 + kTile - 1 + off >= ow));
      need_mask = CAUSAL && (s0 + off < ow + 31);
    }

    if (active) {
      USP_LDS const char* x1 = smem + buf * BUFB;
      USP_LDS const char* x2 = x1 + TILEB;
      u32x4 pk_ds[4], pk_p[MODE == 1 ? 4 : 1];
#pragma unroll
      for (int n32 = 0; n32 < 2; ++n32) {
        f32x16 s, tt;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; tt[r] = 0.f; }
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          const int a = n32 * 32 * ROWB + rd_row + (((2 * kt) ^ rd_x) * 16);
          const u32x4 f1 = *(USP_LDS const u32x4*)(x1 + a);
          const u32x4 f2 = *(USP_LDS const u32x4*)(x2 + a);
          s = E::mfma(f1, r1[kt], s);
          tt = E::mfma(f2, r2[kt], tt);
        }
        // streamed row of register r: s0 + 32 n32 + 8 (r>>2) + 4 hi + (r&3)
        const int sr0 = s0 + 32 * n32 + 4 * hi;
        if (need_mask) {
          if (MODE == 0) {
            int klim = p.Sk - 1;
            if (CAUSAL) klim = orow + off < klim ? orow + off : klim;
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (sr0 + (r & 3) + 8 * (r >> 2) > klim) s[r] = USP_NEG_INF;
          } else {
            // query row i sees key j iff j <= i + off
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (orow > sr0 + (r & 3) + 8 * (r >> 2) + off) s[r] = USP_NEG_INF;
          }
        }
        float pr[16], dsr[16];
        if (MODE == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            pr[r] = fast_exp2(__builtin_fmaf(s[r], c, -lse2_l));
            dsr[r] = pr[r] * (tt[r] - delta_l);
          }
        } else {
          USP_LDS const char* stat = x1 + 2 * TILEB + (32 * n32 + 4 * hi) * 4;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 l4 = *(USP_LDS const f32x4*)(stat + 32 * g4);
            const f32x4 d4 = *(USP_LDS const f32x4*)(stat + 4 * kTile + 32 * g4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r = 4 * g4 + j;
              pr[r] = fast_exp2(__builtin_fmaf(s[r], c, -l4[j]));
              dsr[r] = pr[r] * (tt[r] - d4[j]);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pk_ds[2 * n32][j] = E::pack2(dsr[2 * j], dsr[2 * j + 1]);
          pk_ds[2 * n32 + 1][j] = E::pack2(dsr[8 + 2 * j], dsr[8 + 2 * j + 1]);
          if (MODE == 1) {
            pk_p[2 * n32][j] = E::pack2(pr[2 * j], pr[2 * j + 1]);
            pk_p[2 * n32 + 1][j] = E::pack2(pr[8 + 2 * j], pr[8 + 2 * j + 1]);
          }
        }
      }
      // gradient MFMAs: acc1^T += X1^T dS ; (MODE 1) acc2^T += X2^T P
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int dj = 0; dj < NDJ; ++dj) {
          const int base = ks * 16 * ROWB;
          const u32x2 a0 = lds_read_tr16(x1 + base + tr_addr[dj][0]);
          const u32x2 a1 = lds_read_tr16(x1 + base + tr_addr[dj][1]);
          const u32x4 xa = {a0[0], a0[1], a1[0], a1[1]};
          acc1[dj] = E::mfma(xa, pk_ds[ks], acc1[dj]);
          if (MODE == 1) {
            const u32x2 b0 = lds_read_tr16(x2 + base + tr_addr[dj][0]);
            const u32x2 b1 = lds_read_tr16(x2 + base + tr_addr[dj][1]);
            const u32x4 xb = {b0[0], b0[1], b1[0], b1[1]};
            acc2[dj] = E::mfma(xb, pk_p[ks], acc2[dj]);
          }
        }
      }
    }

    if (it + 1 < n_iter) stage_store(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: fp32 store / accumulate ------------------------------------------------------------
  if (orow < own_len) {
    float* o1; float* o2 = nullptr;
    int acc_f1, acc_f2 = 0;
    if (MODE == 0) {
      o1 = p.dq + b * p.dq_sb + (int64_t)orow * p.dq_ss + h0 * p.dq_sh; acc_f1 = p.accum_dq;
    } else {
      o1 = p.dk + b * p.dk_sb + (int64_t)orow * p.dk_ss + hkv * p.dk_sh; acc_f1 = p.accum_dk;
      o2 = p.dv + b * p.dv_sb + (int64_t)orow * p.dv_ss + hkv * p.dv_sh; acc_f2 = p.accum_dv;
    }
#pragma unroll
    for (int dj = 0; dj < NDJ; ++dj)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int d0 = 32 * dj + 8 * g4 + 4 * hi;
        f32x4 v1 = {acc1[dj][4 * g4] * p.scale, acc1[dj][4 * g4 + 1] * p.scale,
                    acc1[dj][4 * g4 + 2] * p.scale, acc1[dj][4 * g4 + 3] * p.scale};
        if (acc_f1) v1 += *(const f32x4*)(o1 + d0);
        *(f32x4*)(o1 + d0) = v1;
        if (MODE == 1) {
          f32x4 v2 = {acc2[dj][4 * g4], acc2[dj][4 * g4 + 1], acc2[dj][4 * g4 + 2],
                      acc2[dj][4 * g4 + 3]};
          if (acc_f2) v2 += *(const f32x4*)(o2 + d0);
          *(f32x4*)(o2 + d0) = v2;
        }
      }
  }
}

template <int D, int DT>
static int launch_bwd(BwdParams p, bool causal, hipStream_t st) {
  constexpr size_t lds0 = 2 * (2 * kTile * D * 2);
  constexpr size_t lds1 = 2 * (2 * kTile * D * 2 + 2 * kTile * 4);
  // dK,dV
  p.nblk = (p.Sk + 127) / 128;
  int grid = p.B * p.Hkv * p.nblk;
  if (causal)
    hipLaunchKernelGGL((flash_bwd_kernel<D, DT, true, 1>), dim3(grid), dim3(256), lds1, st, p);
  else
    hipLaunchKernelGGL((flash_bwd_kernel<D, DT, false, 1>), dim3(grid), dim3(256), lds1, st, p);
  if (hipGetLastError() != hipSuccess) return USP_ELAUNCH;
  // dQ
  p.nblk = (p.Sq + 255) / 256;
  grid = p.B * p.Hq * p.nblk;
  if (causal)
    hipLaunchKernelGGL((flash_bwd_kernel<D, DT, true, 0>), dim3(grid), dim3(512), lds0, st, p);
  else
    hipLaunchKernelGGL((flash_bwd_kernel<D, DT, false, 0>), dim3(grid), dim3(512), lds0, st, p);
  return hipGetLastError() == hipSuccess ? USP_OK : USP_ELAUNCH;
}

static bool ok16(const usp_tensor& t, int esize) {
  const int m = 16 / esize;
  return t.ptr && (reinterpret_cast<uintptr_t>(t.ptr) & 15) == 0 && t.stride_b % m == 0 &&
         t.stride_s % m == 0 && t.stride_h % m == 0;
}

}  // namespace usp

extern "C" int usp_flash_bwd(const usp_bwd_args* a, void* stream) {
  using namespace usp;
  if (!a || !a->lse || !a->delta) return USP_EINVAL;
  if (a->dtype != USP_BF16 && a->dtype != USP_FP16) return USP_EINVAL;
  if (a->B <= 0 || a->Sq <= 0 || a->Sk <= 0 || a->Hq <= 0 || a->Hkv <= 0) return USP_EINVAL;
  if (!(a->softmax_scale > 0.f)) return USP_EINVAL;
  if (a->D != 32 && a->D != 64 && a->D != 128) return USP_EUNSUPPORTED;
  if (a->Hq % a->Hkv != 0) return USP_EUNSUPPORTED;
  if (!a->dout.ptr || !a->q.ptr || !a->k.ptr || !a->v.ptr || !a->dq.ptr || !a->dk.ptr || !a->dv.ptr)
    return USP_EINVAL;
  if (!ok16(a->dout, 2) || !ok16(a->q, 2) || !ok16(a->k, 2) || !ok16(a->v, 2) || !ok16(a->dq, 4) ||
      !ok16(a->dk, 4) || !ok16(a->dv, 4))
    return USP_EUNSUPPORTED;
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
  p.causal_off = a->Sk - a->Sq;
  p.scale = a->softmax_scale;
  p.scale_log2 = a->softmax_scale * kLog2e;
  p.accum_dq = a->accum_dq ? 1 : 0; p.accum_dk = a->accum_dk ? 1 : 0; p.accum_dv = a->accum_dv ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  const bool causal = a->causal != 0;
  switch (a->D * 2 + a->dtype) {
    case 64: return launch_bwd<32, 0>(p, causal, st);
    case 65: return launch_bwd<32, 1>(p, causal, st);
    case 128: return launch_bwd<64, 0>(p, causal, st);
    case 129: return launch_bwd<64, 1>(p, causal, st);
    case 256: return launch_bwd<128, 0>(p, causal, st);
    case 257: return launch_bwd<128, 1>(p, causal, st);
  }
  return USP_EUNSUPPORTED;
}
