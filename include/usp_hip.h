/*
 * usp_hip.h -- C ABI of libusp_hip.so, the MI355X (gfx950) device library behind the
 * Unified-Sequence-Parallel attention path.
 *
 * The reference (feifeibear/long-context-attention, "yunchang" v0.6.4) is pure Python and has no
 * FFI of its own: its device arithmetic lives in third-party ops reached through the selector
 * seam `select_flash_attn_impl` (yunchang/kernels/__init__.py:63-65).  Each entry point below
 * replaces one of those call sites (cited per function); the Python binding a maintainer adds is
 * a ctypes stub, shown in INTEGRATION.md and implemented in long-context-attention_amd/_C.py.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless stated;
 *   - tensors are (batch, seq, head, dim) with dim contiguous; strides are in ELEMENTS;
 *   - nothing allocates, nothing synchronises, nothing touches any stream but `stream`
 *     (a hipStream_t passed as void*; NULL = the legacy default stream);
 *   - thread-safe and re-entrant: no mutable globals;
 *   - return 0 on success, a negative USP_E* code otherwise (the launch is then not issued);
 *     usp_strerror() maps codes to text.
 */
#ifndef USP_HIP_H
#define USP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define USP_ABI_VERSION 7

enum { USP_BF16 = 0, USP_FP16 = 1 };

enum {
  USP_OK = 0,
  USP_EINVAL = -1,      /* bad argument (null pointer, non-positive size, bad dtype) */
  USP_EUNSUPPORTED = -2,/* head_dim not in {32, 64, 128}, Hq % Hkv != 0, misaligned pointer/stride */
  USP_ELAUNCH = -3      /* hipLaunchKernel reported an error */
};

/* A (batch, seq, head, dim) view: element pointer + element strides; dim stride is 1. */
/* Launch flags (usp_fwd_args.flags / usp_bwd_args.flags).
 * USP_LAUNCH_INTERLEAVE: other kernels -- RCCL's send/recv and all-to-all kernels on another stream -- must
 * be able to start WHILE this launch runs.  By default the flash kernels are persistent: one workgroup per
 * CU for the whole launch, each holding the CU's entire register file, so nothing else can be scheduled
 * until the launch ends (fastest when the GPU does nothing else: +4 % forward).  With this flag one
 * workgroup per work item is launched instead; CUs free up every few microseconds and a collective queued
 * on another stream gets its workgroups resident at once.  Use it whenever a transfer is meant to overlap
 * the kernel (ring steps, pipelined Ulysses exchange). */
#define USP_LAUNCH_INTERLEAVE 1
/* USP_ATTN_WINDOW (ABI v5): the window_left / window_right fields are valid (without the bit they are ignored, so a
 * zero-initialised struct means "no window", not "window (0, 0)"). */
#define USP_ATTN_WINDOW 2
/* Kernel-family selectors (ABI v6; usp_flash_fwd / usp_flash_bwd).  The library holds two families of flash kernels: the
 * one-wave-per-SIMD family (a wave owns 64 rows / keys and its SIMD's whole register file: usp_flash_fwd64.hip,
 * usp_flash_bwd64.hip, usp_flash_bwd_dq64.hip) and the two-waves-per-SIMD family (32 rows per wave, 8 or 4 waves:
 * usp_flash_fwd.hip, usp_flash_bwd.hip), and picks per launch.  With one of these bits the caller picks -- per CALL, not
 * per process -- which is what lets a parity test pin the kernel it means to test, and a bench time both families on
 * the same box:
 *   USP_FORCE_ROW64   every flash kernel of the call must come from the 64-row family; a call that family does not
 *                     serve (head dim != 128, packed batch, window) returns USP_EUNSUPPORTED and launches nothing
 *                     (unforced, the library picks the 64-row forward K split only for long cuts);
 *   USP_FORCE_WAVE32  every flash kernel of the call comes from the 32-rows-per-wave family.
 * Both bits at once: USP_EINVAL.  usp_last_launch_kinds() reports what a call actually launched. */
#define USP_FORCE_ROW64 4
#define USP_FORCE_WAVE32 8
/* usp_flash_bwd only (ABI v6): issue only one of the backward's two launches -- the dK/dV launch (+ its head / cut reduce)
 * or the dQ launch (+ its cut reduce).  The two are independent given (lse, delta); a caller may order them around its
 * transfers (dK/dV travel the ring, dQ stays), and a bench can time each kernel alone with stream events.  The skipped
 * launch's outputs are not touched.  Both bits at once: USP_EINVAL. */
#define USP_BWD_SKIP_DQ 16
#define USP_BWD_SKIP_DKDV 32

typedef struct usp_tensor {
  void* ptr;
  int64_t stride_b, stride_s, stride_h;
} usp_tensor;

/* ----------------------------------------------------------------------------------------------
 * Blockwise flash-attention forward, with the ring LSE-merge fused into the epilogue.
 *
 * Replaces the `fwd-only` callable: pytorch_attn_forward(op_type="efficient")
 * (yunchang/kernels/attention.py:44-136, aten op at :76-86) / flash_attn_forward (:165-202), as
 * called by the ring schedules (yunchang/ring/zigzag_ring_flash_attn.py:29-43,
 * yunchang/ring/ring_flash_attn.py:36-48), AND the update_out_and_lse that follows each call
 * (yunchang/ring/utils.py:10-51; slice form used at zigzag_ring_flash_attn.py:61-67).
 *
 * Computes, for q (B,Sq,Hq,D), k/v (B,Sk,Hkv,D) [GQA: q head i uses kv head i / (Hq/Hkv)]:
 *     S = q k^T * softmax_scale ; causal: key j visible to query i iff j <= i + (Sk - Sq)
 *     blk_lse = logsumexp_j S ;  blk_out = softmax(S) v          (fp32 accumulate)
 * then, per query row i of this call:
 *     merge_in != 0 : (o, lse) = merge((acc[i], lse[i]), (blk_out, blk_lse))   [utils.py:25-26]
 *     merge_in == 0 : (o, lse) = (blk_out, blk_lse)
 *     lse[i] <- lse                                   (fp32, natural log; -inf for an empty row)
 *     i in [final_begin, final_end) : out[i] <- o rounded to `dtype`   (needs out.ptr != NULL)
 *     otherwise                      : acc[i] <- o  (fp32)             (needs acc.ptr != NULL)
 * `lse` is (B,Hq,Sq) with seq stride 1.  A single non-ring call uses merge_in=0,
 * final_begin=0, final_end=Sq, acc.ptr=NULL.
 *
 * Packed variable-length batches (replaces _flash_attn_varlen_forward as called at
 * yunchang/ring/zigzag_ring_flash_attn_varlen.py:88-112 and ring_flash_attn_varlen.py:56-71, the
 * k[half_index0] / q[half_index1] gathers at zigzag_ring_flash_attn_varlen.py:127-140, and the LSE
 * (un)flatten of yunchang/ring/utils.py:96-117): seq_q / seq_k point to B (first_row, rows) int32
 * pairs in DEVICE memory.  q/out/acc are (T,Hq,D) and k/v (T',Hkv,D) row-major token tensors
 * (stride_b ignored); sequence b attends rows [first, first+rows) of each side, causal alignment per
 * sequence; lse is (Hq,T): lse[h*lse_stride_h + row].  Sq / Sk are the MAXIMUM rows over the batch
 * (they size the launch).  Sequences with 0 query rows are skipped; a sequence with 0 key rows gives
 * blk_out = 0, blk_lse = -inf.  A (first,rows) pair may address any sub-range of the token tensors,
 * e.g. the front or back half of every sequence -- no gather copies are needed.  In this mode
 * final_begin / final_end count HALF sequences (0, 1 or 2): rows [final_begin*rows/2,
 * final_end*rows/2) of every sequence are final (2 = to the end).
 * `sched` (packed mode only, optional): 16 int32 of device memory, ZEROED ONCE by the caller.  With it
 * the workgroups pull work items from a dynamic queue (heaviest first, empty items of short sequences
 * skipped), which is what balances batches of unequal sequences; without it they walk static lists,
 * which is only balanced when all sequences are equally long.  The kernels leave the block zeroed;
 * launches that may run concurrently (different streams) must not share one block.
 * -------------------------------------------------------------------------------------------- */
typedef struct usp_fwd_args {
  int32_t dtype;                 /* USP_BF16 | USP_FP16: element type of q,k,v,out */
  int32_t B, Sq, Sk, Hq, Hkv, D;
  int32_t causal;                /* 0 | 1 (bottom-right aligned) */
  float softmax_scale;           /* > 0 */
  usp_tensor q, k, v;            /* inputs */
  usp_tensor out;                /* 16-bit output rows (final rows only); ptr may be NULL */
  usp_tensor acc;                /* fp32 running output; ptr may be NULL */
  float* lse;                    /* fp32 running / final logsumexp */
  int64_t lse_stride_b, lse_stride_h;
  int32_t merge_in;              /* 0 | 1 */
  int32_t final_begin, final_end;/* row range of this call whose result is final */
  const int32_t* seq_q;          /* packed variable-length batch (both NULL = dense), see below */
  const int32_t* seq_k;
  int32_t* sched;                /* packed mode, optional: 64-byte device control block, see below */
  int32_t flags;                 /* USP_LAUNCH_* bits */
  int32_t k_splits;              /* dense mode, optional: cut every query tile's keys into this many work items
                                    (2..8; 0 or 1 = off); needs `workspace`, see usp_flash_fwd_workspace_bytes() */
  void* workspace;               /* device scratch of usp_flash_fwd_workspace_bytes(args, k_splits) bytes, 16-byte
                                    aligned; NULL = no split */
  int32_t window_left, window_right; /* read only with USP_ATTN_WINDOW in `flags` (ABI v5): sliding-window (local)
                                    attention as flash-attn defines it (kernels/attention.py:165-202 passes
                                    `window_size` through): query row i sees key j iff
                                    i + (Sk - Sq) - window_left <= j <= i + (Sk - Sq) + window_right; a negative value =
                                    unbounded on that side; `causal` caps window_right at 0 */
} usp_fwd_args;

int usp_flash_fwd(const usp_fwd_args* args, void* stream);

/* K split (dense launches): a launch with few (batch, head, 256-row tile) work items cannot fill 256 CUs, and a causal
 * one lasts as long as its heaviest item (measured on MI355X, B1 S16384 D128 causal: 2 heads 548 TFLOP/s, 4 heads 891,
 * 8 heads 1133; the same 2 / 4 heads cut in two along K: 833-857 / 1093 -- profiles/r02_kbench_split.log).  With
 * k_splits = n and a workspace every item becomes n items over equal runs of the K tiles it sees; each writes a
 * normalised fp32 partial + LSE to the workspace and a second, HBM-bound launch on the same stream combines them [and
 * the running result when merge_in] into exactly the outputs described above (results equal up to fp32 summation
 * order).  Returns the bytes needed: n * (B*Sq*Hq*D + B*Hq*Sq) * 4; 0 for n <= 1 or a packed batch. */
int64_t usp_flash_fwd_workspace_bytes(const usp_fwd_args* args, int32_t k_splits);

/* ----------------------------------------------------------------------------------------------
 * Blockwise flash-attention backward.
 *
 * Replaces the `bwd-only` callable flash_attn_backward (yunchang/kernels/attention.py:205-250;
 * the TORCH_* variant pytorch_attn_backward :138-159 raises), as called at
 * zigzag_ring_flash_attn.py:115-137 / ring_flash_attn.py:102-122, AND the fp32 accumulation the
 * ring schedule performs on its results (zigzag_ring_flash_attn.py:147-170).
 *
 * Inputs: dout,q (B,Sq,Hq,D); k,v (B,Sk,Hkv,D); lse (B,Hq,Sq) = the GLOBAL logsumexp of the rows
 * (not this block's); delta (B,Hq,Sq) = rowsum(dout * out_global) from usp_bwd_delta().
 *     P = exp(S - lse) ; dV = P^T dO ; dP = dO V^T ; dS = P * (dP - delta) * scale
 *     dQ = dS K ; dK = dS^T Q                  (GQA: dK,dV summed over the Hq/Hkv query heads)
 * Outputs are fp32 (B,S,H,D) tensors: dq (+)= dQ, dk (+)= dK, dv (+)= dV, where "+=" is used when
 * the matching accum_* flag is non-zero and "=" otherwise.  Deterministic (no atomics).
 * `workspace` (optional) enables the GQA head split described at usp_flash_bwd_workspace_bytes().
 *
 * Packed variable-length batches (replaces _flash_attn_varlen_backward as called at
 * yunchang/ring/zigzag_ring_flash_attn_varlen.py:208-231 / ring_flash_attn_varlen.py:125-147, the
 * half-index gathers / scatters around it (:246-264) and get_half_lse (:45-58)): seq_q / seq_k as in
 * usp_fwd_args -- B (first_row, rows) pairs in device memory; dout/q/dq/dq16 are (T,Hq,D) token
 * tensors addressed through seq_q, k/v/dk/dv/dk16/dv16 (T',Hkv,D) through seq_k, lse/delta are
 * (Hq,T); Sq / Sk are the maximum rows over the batch.  Gradient rows outside every sequence's
 * range are not touched; sequences with 0 rows on either side are skipped entirely.
 * -------------------------------------------------------------------------------------------- */
typedef struct usp_bwd_args {
  int32_t dtype;
  int32_t B, Sq, Sk, Hq, Hkv, D;
  int32_t causal;
  float softmax_scale;
  usp_tensor dout, q, k, v;      /* 16-bit inputs */
  const float* lse;              /* (B,Hq,Sq), seq stride 1 */
  const float* delta;            /* (B,Hq,Sq), seq stride 1 */
  int64_t lse_stride_b, lse_stride_h;
  int64_t delta_stride_b, delta_stride_h;
  usp_tensor dq, dk, dv;         /* fp32 outputs / accumulators */
  int32_t accum_dq, accum_dk, accum_dv;
  usp_tensor dq16, dk16, dv16;   /* optional 16-bit FINAL outputs (ptr may be NULL): when set, the
                                    result ((accum ? X : 0) + block) is rounded to `dtype` and stored
                                    there instead of being written back to the fp32 tensor X */
  void* workspace;               /* optional scratch (device, 16-byte aligned), see below; may be NULL */
  int64_t workspace_bytes;
  const int32_t* seq_q;          /* packed variable-length batch (both NULL = dense), see below */
  const int32_t* seq_k;
  int64_t total_k;               /* packed mode: rows of the k/v/dk/dv token tensors (sizes the workspace) */
  int32_t* sched;                /* packed mode, optional: 64-byte device control block (as usp_fwd_args) */
  int32_t flags;                 /* USP_LAUNCH_* / USP_ATTN_* bits */
  int32_t dq_splits;             /* dense mode, optional (ABI v5): cut every (head, 256-row query block) item of the dQ
                                    launch into this many items along the keys it sees (2..8; 0 / 1 = off) */
  int32_t dkdv_splits;           /* ... and every (query head, 128-key block) item of the dK/dV launch along the query
                                    rows that see it.  Both need `workspace`, see usp_flash_bwd_workspace_bytes() */
  int32_t window_left, window_right; /* as usp_fwd_args; read only with USP_ATTN_WINDOW in `flags` */
  int32_t dkdv_heads;            /* GQA, dense or packed (ABI v7): query heads of a KV group that ONE dK/dV work item streams
                                    into its accumulators -- a divisor of Hq/Hkv (else USP_EINVAL).  Hq/Hkv = the whole group
                                    inside the workgroup (no per-head partials, no reduce launch); 1 = one item per query
                                    head; 0 = the library decides (see usp_flash_bwd_workspace_bytes()) */
} usp_bwd_args;

int usp_flash_bwd(const usp_bwd_args* args, void* stream);

/* Bytes of scratch with which the backward launches get more, smaller work items (fp32 partials + one deterministic,
 * HBM-bound reduce launch each; results identical up to fp32 summation order):
 *   - GQA (Hq > Hkv): a dK/dV work item streams dkdv_heads query heads of its KV group (ABI v7; rounds 1-5: one or all).
 *     With fewer than Hq/Hkv heads per item there are (Hq/Hkv)/dkdv_heads times more items -- what balances the causal
 *     triangle when B*Hkv*ceil(Sk/128) is small -- and as many fp32 partial slabs.  dkdv_heads = 0: the largest divisor of
 *     Hq/Hkv up to 2 that leaves two work items per CU for causal / windowed launches (more heads per item let the concurrent
 *     workgroups' Q / dO streams drift out of one L2), up to 4 and one item per CU for full ones -- measured:
 *     profiles/r06_gqa_loop.txt; packed batches: 1;
 *   - dkdv_splits = n (ABI v5): every such item is cut into n items over equal runs of the query tiles that see its keys;
 *   - dq_splits = n (ABI v5): every (head, 256-row query block) item of the dQ launch is cut into n items over equal runs
 *     of the key tiles it sees -- for launches with few heads (a small head group, a high Ulysses degree), where a causal
 *     launch otherwise lasts as long as its heaviest item.
 * = 2 * ((Hq/Hkv)/dkdv_heads) * max(1, dkdv_splits) * rows_k * Hkv * D * 4  (0 when that is a single slab)
 *   + dq_splits * B * Sq * Hq * D * 4                           (0 when dq_splits <= 1); the cuts are ignored for packed
 * batches.  Passing less (or NULL) is valid and selects the whole group per work item without cuts. */
int64_t usp_flash_bwd_workspace_bytes(const usp_bwd_args* args);

/* delta[b,h,s] = sum_d dout[b,s,h,d] * out[b,s,h,d]   (fp32; delta is (B,H,S), seq stride 1).
 * The "softmax_d" term flash-attn's backward derives from `out` (its 5th positional argument,
 * kernels/attention.py:205). */
int usp_bwd_delta(int32_t dtype, int32_t B, int32_t S, int32_t H, int32_t D,
                  const usp_tensor* dout, const usp_tensor* out,
                  float* delta, int64_t delta_stride_b, int64_t delta_stride_h, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Stand-alone LSE merge: update_out_and_lse (yunchang/ring/utils.py:10-51) as one kernel.
 *   acc (B,S,H,D) fp32, lse (B,H,S) fp32 are updated in place with a block result
 *   blk_out (B,S,H,D) 16-bit, blk_lse (B,H,S) fp32.  first != 0: adopt the block (utils.py:38-42).
 * -------------------------------------------------------------------------------------------- */
int usp_lse_merge(int32_t dtype, int32_t B, int32_t S, int32_t H, int32_t D,
                  const usp_tensor* acc, float* lse, int64_t lse_stride_b, int64_t lse_stride_h,
                  const usp_tensor* blk_out, const float* blk_lse, int64_t blk_lse_stride_b,
                  int64_t blk_lse_stride_h, int32_t first, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Row-gather copy used to build / unpack the Ulysses all-to-all buffers
 * (the two `.contiguous()` transposes of yunchang/comm/all_to_all.py:39-49,62-65,76-84,98-100).
 *   for i0<n0, i1<n1, i2<n2, i3<n3:
 *     dst[i0*ds0 + i1*ds1 + i2*ds2 + i3*ds3 + (0..row_bytes)] = src[i0*ss0 + ... ]
 * Strides in BYTES; row_bytes, all strides and both pointers must be multiples of 16.
 * -------------------------------------------------------------------------------------------- */
int usp_copy_rows(void* dst, const void* src, int64_t row_bytes,
                  int64_t n0, int64_t n1, int64_t n2, int64_t n3,
                  int64_t ds0, int64_t ds1, int64_t ds2, int64_t ds3,
                  int64_t ss0, int64_t ss1, int64_t ss2, int64_t ss3, void* stream);

/* dst16[i] = round(src32[i]) for i < n, `rows` rows of `n` contiguous elements with row strides
 * (elements) -- the `.to(q.dtype)` casts at zigzag_ring_flash_attn.py:74,183. */
int usp_cast_from_f32(int32_t dtype, void* dst, int64_t dst_row_stride, const float* src,
                      int64_t src_row_stride, int64_t rows, int64_t n, void* stream);

/* dst[r, i] = a[r, i] + b[r, i]  (fp32; any of the three may alias) -- the dk/dv accumulator adds
 * at zigzag_ring_flash_attn.py:161-170 / ring_flash_attn.py:130-131. */
int usp_add_f32(float* dst, int64_t dst_row_stride, const float* a, int64_t a_row_stride,
                const float* b, int64_t b_row_stride, int64_t rows, int64_t n, void* stream);

/* What the LAST usp_flash_fwd / usp_flash_bwd call of the CALLING THREAD launched: a mask of USP_KIND_* bits (0 before
 * the first call and after a call that returned an error before launching).  Thread-local: no state is shared between
 * threads, the entry points stay re-entrant.  For tests ("assert the 64-row forward ran") and bench labels. */
enum {
  USP_KIND_FWD_ROW64 = 1,       /* flash_fwd64_kernel        (4 waves x 64 rows) */
  USP_KIND_FWD_WAVE8 = 2,       /* flash_fwd_kernel, 8 waves (256 rows per item) */
  USP_KIND_FWD_WAVE4 = 4,       /* flash_fwd_kernel, 4 waves (128 rows per item) */
  USP_KIND_FWD_SPLIT_MERGE = 8, /* split_merge_kernel behind a K-split launch */
  USP_KIND_DKDV_ROW64 = 16,     /* flash_bwd_dkdv64_kernel */
  USP_KIND_DKDV_WAVE8 = 32,     /* flash_bwd_dkdv_kernel */
  USP_KIND_DQ_ROW64 = 64,       /* flash_bwd_dq64_kernel */
  USP_KIND_DQ_WAVE8 = 128,      /* flash_bwd_kernel (dQ) */
  USP_KIND_REDUCE_HEADS = 256,  /* reduce_heads_kernel (GQA head split / dK,dV cuts) */
  USP_KIND_REDUCE_CUTS = 512    /* reduce_cuts_kernel (dQ key cuts) */
};
int usp_last_launch_kinds(void);

/* Diagnostic (ABI v6), not a replacement of any reference call site: what the matrix pipe of THIS part sustains on the
 * caller's operands -- an MFMA-only loop (v_mfma_f32_32x32x16_bf16, no LDS / VALU / memory traffic inside) on one workgroup
 * per CU, `waves_per_simd` (1 | 2) waves per SIMD, `iters` passes of 64 MFMAs per wave.  `operands`: >= 64 KiB of device
 * memory holding bf16 values (N(0,1) as the bench's inputs, or zeros: the part clocks by power and MFMA power depends on
 * the operand bit patterns).  FLOPs of a launch = CUs * 4 * waves_per_simd * iters * 64 * 32768.  `sink`: >= 512 floats of
 * device scratch.  `clocks` (optional, 2 x uint64 device): elapsed s_memtime (shader clock) and s_memrealtime (100 MHz)
 * ticks of one wave's loop -> sustained clock.  bench.py reports a flash kernel's rate against this ceiling beside its
 * fraction of the nominal 2.5 PFLOP/s. */
int usp_mfma_probe(const void* operands, int64_t operand_bytes, int32_t iters, int32_t waves_per_simd,
                   float* sink, uint64_t* clocks, void* stream);

int usp_abi_version(void);
const char* usp_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif /* USP_HIP_H */
