/*
 * attn_oracle.c -- plain-C CPU restatement of the block attention contract on the USP hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/usp_oracle.py header): linked by the native kernel harness
 * (long-context-attention_amd/csrc/tools/kbench.cpp) and loaded by tests/ and bench.py's
 * cpu_baseline leg.  Never linked into libusp_hip.so.
 *
 * Follows test/test_utils.py:43-130 (attention_ref: fp32/fp64 upcast, GQA by q head i -> kv head
 * i / g, bottom-right aligned causal mask :35-36) for the forward, and the flash-attn backward
 * contract of yunchang/kernels/attention.py:205-250 (global out / lse) for the backward
 * (formulas: SURVEY.md Appendix A).  Inputs are float arrays in (B,S,H,D) layout, contiguous;
 * all accumulation in double.  Parity pinned through tests/test_oracle_golden.py (the numpy
 * oracle, which is pinned to reference runs) -- tests/test_c_oracle.py checks C == numpy.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IDX4(b, s, h, d, S, H, D) ((((int64_t)(b) * (S) + (s)) * (H) + (h)) * (D) + (d))

/* out (B,Sq,Hq,D), lse (B,Hq,Sq) */
void usp_oracle_attn_fwd(const float* q, const float* k, const float* v, int B, int Sq, int Sk,
                         int Hq, int Hkv, int D, float scale, int causal, float* out, float* lse) {
  const int g = Hq / Hkv;
  const int off = Sk - Sq;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
  for (int b = 0; b < B; ++b) {
    for (int h = 0; h < Hq; ++h) {
      double* p = (double*)malloc(sizeof(double) * (size_t)Sk);
      double* acc = (double*)malloc(sizeof(double) * (size_t)D);
      const int hk = h / g;
      for (int i = 0; i < Sq; ++i) {
        int kend = Sk;
        if (causal) { kend = i + off + 1; if (kend > Sk) kend = Sk; }
        const float* qi = q + IDX4(b, i, h, 0, Sq, Hq, D);
        float* oi = out + IDX4(b, i, h, 0, Sq, Hq, D);
        if (kend <= 0) {
          memset(oi, 0, sizeof(float) * (size_t)D);
          lse[((int64_t)b * Hq + h) * Sq + i] = -INFINITY;
          continue;
        }
        double m = -INFINITY;
        for (int j = 0; j < kend; ++j) {
          const float* kj = k + IDX4(b, j, hk, 0, Sk, Hkv, D);
          double s = 0;
          for (int d = 0; d < D; ++d) s += (double)qi[d] * kj[d];
          s *= scale;
          p[j] = s;
          if (s > m) m = s;
        }
        double l = 0;
        for (int j = 0; j < kend; ++j) { p[j] = exp(p[j] - m); l += p[j]; }
        for (int d = 0; d < D; ++d) acc[d] = 0;
        for (int j = 0; j < kend; ++j) {
          const float* vj = v + IDX4(b, j, hk, 0, Sk, Hkv, D);
          const double pj = p[j];
          for (int d = 0; d < D; ++d) acc[d] += pj * vj[d];
        }
        for (int d = 0; d < D; ++d) oi[d] = (float)(acc[d] / l);
        lse[((int64_t)b * Hq + h) * Sq + i] = (float)(m + log(l));
      }
      free(p);
      free(acc);
    }
  }
}

/* dq (B,Sq,Hq,D), dk/dv (B,Sk,Hkv,D); out/lse are the GLOBAL rows' values. */
void usp_oracle_attn_bwd(const float* dout, const float* q, const float* k, const float* v,
                         const float* out, const float* lse, int B, int Sq, int Sk, int Hq, int Hkv,
                         int D, float scale, int causal, float* dq, float* dk, float* dv) {
  const int g = Hq / Hkv;
  const int off = Sk - Sq;
  memset(dq, 0, sizeof(float) * (size_t)B * Sq * Hq * D);
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
  for (int b = 0; b < B; ++b) {
    for (int hk = 0; hk < Hkv; ++hk) {
      double* dka = (double*)calloc((size_t)Sk * D, sizeof(double));
      double* dva = (double*)calloc((size_t)Sk * D, sizeof(double));
      double* dqa = (double*)malloc(sizeof(double) * (size_t)D);
      for (int gi = 0; gi < g; ++gi) {
        const int h = hk * g + gi;
        for (int i = 0; i < Sq; ++i) {
          int kend = Sk;
          if (causal) { kend = i + off + 1; if (kend > Sk) kend = Sk; }
          const float* qi = q + IDX4(b, i, h, 0, Sq, Hq, D);
          const float* doi = dout + IDX4(b, i, h, 0, Sq, Hq, D);
          const float* oi = out + IDX4(b, i, h, 0, Sq, Hq, D);
          const double L = lse[((int64_t)b * Hq + h) * Sq + i];
          double delta = 0;
          for (int d = 0; d < D; ++d) delta += (double)doi[d] * oi[d];
          for (int d = 0; d < D; ++d) dqa[d] = 0;
          if (isfinite(L)) {
            for (int j = 0; j < kend; ++j) {
              const float* kj = k + IDX4(b, j, hk, 0, Sk, Hkv, D);
              const float* vj = v + IDX4(b, j, hk, 0, Sk, Hkv, D);
              double s = 0, dp = 0;
              for (int d = 0; d < D; ++d) { s += (double)qi[d] * kj[d]; dp += (double)doi[d] * vj[d]; }
              const double p = exp(s * scale - L);
              const double ds = p * (dp - delta) * scale;
              double* dkj = dka + (size_t)j * D;
              double* dvj = dva + (size_t)j * D;
              for (int d = 0; d < D; ++d) {
                dqa[d] += ds * kj[d];
                dkj[d] += ds * qi[d];
                dvj[d] += p * doi[d];
              }
            }
          }
          float* dqi = dq + IDX4(b, i, h, 0, Sq, Hq, D);
          for (int d = 0; d < D; ++d) dqi[d] = (float)dqa[d];
        }
      }
      for (int j = 0; j < Sk; ++j)
        for (int d = 0; d < D; ++d) {
          dk[IDX4(b, j, hk, d, Sk, Hkv, D)] = (float)dka[(size_t)j * D + d];
          dv[IDX4(b, j, hk, d, Sk, Hkv, D)] = (float)dva[(size_t)j * D + d];
        }
      free(dka); free(dva); free(dqa);
    }
  }
}

/* update_out_and_lse (yunchang/ring/utils.py:25-26) on fp32 arrays; out (B,S,H,D), lse (B,H,S). */
void usp_oracle_merge(float* out, float* lse, const float* blk_out, const float* blk_lse, int B, int S,
                      int H, int D) {
  for (int b = 0; b < B; ++b)
    for (int s = 0; s < S; ++s)
      for (int h = 0; h < H; ++h) {
        const int64_t li = ((int64_t)b * H + h) * S + s;
        const double o = lse[li], n = blk_lse[li];
        double sig, nl;
        if (isinf(o) && isinf(n) && o < 0 && n < 0) { sig = 0; nl = -INFINITY; }
        else {
          sig = 1.0 / (1.0 + exp(o - n));                    /* sigmoid(blk_lse - lse) */
          const double mx = o > n ? o : n;
          nl = mx + log(exp(o - mx) + exp(n - mx));          /* lse - logsigmoid(lse - blk_lse) */
        }
        for (int d = 0; d < D; ++d) {
          const int64_t oi = IDX4(b, s, h, d, S, H, D);
          out[oi] = (float)(out[oi] - sig * ((double)out[oi] - blk_out[oi]));
        }
        lse[li] = (float)nl;
      }
}
