"""CPU oracle for the Unified-Sequence-Parallel attention hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this file;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may.  It restates, in plain numpy (float64 by default), the algorithm of the
reference (feifeibear/long-context-attention, "yunchang" v0.6.4).  Every function
cites the reference file:line it follows (paths relative to the reference root).

Parity status: the reference ships NO golden vectors / known-answer tests for this
path.  This oracle is therefore pinned against outputs of the
reference itself, run in the build container on CPU/gloo with the recipe of
SURVEY.md Appendix A; the script is ``tests/golden/make_golden.py`` and the fixtures
are ``tests/golden/*.npz`` (checked by ``tests/test_oracle_golden.py``).

Multi-rank behaviour is *simulated* in one process: a "world" is a python list with
one entry per rank, and collectives are list permutations.  That keeps the oracle
independent of torch.distributed (which the product uses).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

NEG_INF = -np.inf


# ----------------------------------------------------------------------------
# process grid  (yunchang/globals.py:22-81)
# ----------------------------------------------------------------------------
def seq_parallel_groups(ud: int, rd: int, world_size: int, use_ulysses_low: bool = True):
    """Return (ulysses_groups, ring_groups): lists of rank lists.

    Follows set_seq_parallel_pg (yunchang/globals.py:22-81): sp = ud*rd, dp = ws//sp;
    with use_ulysses_low the ulysses ranks are contiguous (:39-51) and ring ranks are
    strided by ud (:53-57); otherwise the roles swap (:59-78).
    """
    sp = ud * rd
    assert world_size % sp == 0, f"world_size {world_size} % sp_degree {sp} == 0"
    dp = world_size // sp
    ulysses, ring = [], []
    for d in range(dp):
        off = d * sp
        if use_ulysses_low:
            for i in range(rd):
                ulysses.append(list(range(i * ud + off, (i + 1) * ud + off)))
            for i in range(ud):
                ring.append(list(range(i + off, sp + off, ud)))
        else:
            for i in range(ud):
                ring.append(list(range(i * rd + off, (i + 1) * rd + off)))
            for i in range(rd):
                ulysses.append(list(range(i + off, sp + off, rd)))
    return ulysses, ring


def group_of(rank: int, groups: Sequence[Sequence[int]]) -> List[int]:
    for g in groups:
        if rank in g:
            return list(g)
    raise ValueError(f"rank {rank} in no group")


# ----------------------------------------------------------------------------
# global -> local shard layouts  (yunchang/comm/extract_local.py:25-49)
# ----------------------------------------------------------------------------
def basic_extract_local(x: np.ndarray, rank: int, world_size: int, *a, **kw) -> np.ndarray:
    """extract_local.py:25-26: contiguous chunk ``rank`` of ``world_size`` along dim 1."""
    return np.array(np.array_split(x, world_size, axis=1)[rank])


def zigzag_extract_local(x: np.ndarray, rank: int, world_size: int, rd: int, ud: int,
                         use_ulysses_low: bool = True) -> np.ndarray:
    """extract_local.py:29-49: 2*rd chunks; ring rank r takes chunks r and 2rd-1-r,
    concatenated, then split ud ways and indexed by the ulysses rank."""
    ulysses, ring = seq_parallel_groups(ud, rd, world_size, use_ulysses_low)
    r_rank = group_of(rank, ring).index(rank)
    u_rank = group_of(rank, ulysses).index(rank)
    chunks = np.array_split(x, 2 * rd, axis=1)
    loc = np.concatenate([chunks[r_rank], chunks[2 * rd - r_rank - 1]], axis=1)
    return np.array(np.array_split(loc, ud, axis=1)[u_rank])


def stripe_extract_local(x: np.ndarray, rank: int, world_size: int, rd: int, ud: int,
                         use_ulysses_low: bool = True) -> np.ndarray:
    """extract_local.py:7-22: token t goes to ring position t % rd ((B, S/rd, rd, ...) ->
    (B, rd, S/rd, ...)), then the sequence is cut into world_size contiguous chunks."""
    B, S = x.shape[:2]
    rest = x.shape[2:]
    y = x.reshape(B, S // rd, rd, -1).transpose(0, 2, 1, 3).reshape(B, S, -1)
    y = np.array_split(y, world_size, axis=1)[rank]
    return np.array(y.reshape((B, S // world_size) + tuple(rest)))


EXTRACT = {"basic": basic_extract_local, "zigzag": zigzag_extract_local, "strip": stripe_extract_local}


# ----------------------------------------------------------------------------
# full attention truth  (test/test_utils.py:43-130 attention_ref)
# ----------------------------------------------------------------------------
def _ein(spec, a, b):
    """np.einsum through BLAS (optimize=True: batched matrix products instead of the scalar loop nest -- 20-50x on
    the test sizes; the same products summed in another order, i.e. equal to ~1e-16 relative in fp64)."""
    return np.einsum(spec, a, b, optimize=True)


def _scores(q, k, scale, causal, window=(-1, -1)):
    """q (B,Sq,Hq,D), k (B,Sk,Hkv,D) -> masked scaled scores (B,Hq,Sq,Sk).

    GQA: q head i uses kv head i // g  (test_utils.py:86-87 ``repeat b s h d -> b s (h g) d``).
    Causal mask is bottom-right aligned: key j visible to query i iff j <= i + Sk - Sq
    (test_utils.py:35-36 with window (-1, 0)).  `window` = (left, right) as test_utils.py:8-40
    construct_local_mask (the mask flash-attn's `window_size` stands for, kernels/attention.py:165-202): key j is
    visible to query i iff  i + Sk - Sq - left <= j <= i + Sk - Sq + right, a negative bound = unbounded on that side;
    causal sets right = 0 (test_utils.py:80-81).
    """
    B, Sq, Hq, D = q.shape
    Sk, Hkv = k.shape[1], k.shape[2]
    g = Hq // Hkv
    kk = np.repeat(k, g, axis=2)
    s = _ein("bthd,bshd->bhts", q * scale, kk)
    left, right = window
    if causal:
        right = 0
    row = np.arange(Sq)[:, None]
    col = np.arange(Sk)[None, :]
    if right >= 0:
        s = np.where(col > row + Sk - Sq + right, NEG_INF, s)
    if left >= 0:
        s = np.where(col < row + Sk - Sq - left, NEG_INF, s)
    return s


def attention_ref(q, k, v, causal=False, softmax_scale=None, dtype=np.float64, window=(-1, -1)):
    """Returns (out (B,Sq,Hq,D), lse (B,Hq,Sq)) computed in ``dtype`` (upcast=True path,
    test_utils.py:83-84).  Fully masked rows give out=0, lse=-inf (test_utils.py:115-117)."""
    q, k, v = (np.asarray(t, dtype=dtype) for t in (q, k, v))
    D = q.shape[-1]
    scale = (1.0 / math.sqrt(D)) if softmax_scale is None else softmax_scale
    g = q.shape[2] // k.shape[2]
    s = _scores(q, k, scale, causal, window)
    m = s.max(axis=-1, keepdims=True)
    m_safe = np.where(np.isfinite(m), m, 0.0)
    p = np.exp(s - m_safe)
    l = p.sum(axis=-1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        lse = (m_safe + np.log(l))[..., 0]
        att = np.where(l > 0, p / l, 0.0)
    vv = np.repeat(v, g, axis=2)
    out = _ein("bhts,bshd->bthd", att, vv)
    return out, lse


# ----------------------------------------------------------------------------
# block kernel contract  (yunchang/kernels/attention.py:44-136 forward;
#                         :205-250 flash_attn_backward argument contract)
# ----------------------------------------------------------------------------
def block_fwd(q, k, v, softmax_scale=None, causal=False, dtype=np.float64):
    """(block_out (B,Sq,Hq,D), block_lse (B,Hq,Sq)).  Same maths as attention_ref; named
    separately because it is the seam ``select_flash_attn_impl(..., 'fwd-only')`` returns
    (kernels/__init__.py:155-157)."""
    return attention_ref(q, k, v, causal=causal, softmax_scale=softmax_scale, dtype=dtype)


def block_bwd(dout, q, k, v, out, lse, softmax_scale=None, causal=False, dtype=np.float64, window=(-1, -1)):
    """Block backward taking the GLOBAL ``out`` rows and GLOBAL ``lse`` (B,Hq,Sq), as the ring
    schedule passes them (zigzag_ring_flash_attn.py:115-137).  Returns (dq, dk, dv).

    S = QK^T*scale (+mask) ; P = exp(S - lse) ; dV = P^T dO ; dP = dO V^T ;
    delta = rowsum(dO * O) ; dS = P * (dP - delta) * scale ; dQ = dS K ; dK = dS^T Q
    (equals autograd through attention_ref when out/lse are the
    block's own)."""
    dout, q, k, v, out = (np.asarray(t, dtype=dtype) for t in (dout, q, k, v, out))
    lse = np.asarray(lse, dtype=dtype)
    B, Sq, Hq, D = q.shape
    Hkv = k.shape[2]
    g = Hq // Hkv
    scale = (1.0 / math.sqrt(D)) if softmax_scale is None else softmax_scale
    s = _scores(q, k, scale, causal, window)               # (B,Hq,Sq,Sk)
    lse_safe = np.where(np.isfinite(lse), lse, 0.0)
    p = np.exp(s - lse_safe[..., None])
    p = np.where(np.isfinite(lse)[..., None], p, 0.0)
    vv = np.repeat(v, g, axis=2)
    kk = np.repeat(k, g, axis=2)
    dv_full = _ein("bhts,bthd->bshd", p, dout)        # (B,Sk,Hq,D)
    dp = _ein("bthd,bshd->bhts", dout, vv)
    delta = np.einsum("bthd,bthd->bht", dout, out)
    ds = p * (dp - delta[..., None]) * scale
    dq = _ein("bhts,bshd->bthd", ds, kk)
    dk_full = _ein("bhts,bthd->bshd", ds, q)
    Sk = k.shape[1]
    dk = dk_full.reshape(B, Sk, Hkv, g, D).sum(axis=3)
    dv = dv_full.reshape(B, Sk, Hkv, g, D).sum(axis=3)
    return dq, dk, dv


# ----------------------------------------------------------------------------
# LSE merge  (yunchang/ring/utils.py:10-51)
# ----------------------------------------------------------------------------
def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _logsigmoid(x):
    return -np.logaddexp(0.0, -x)


def update_out_and_lse(out, lse, block_out, block_lse, row_slice: Optional[slice] = None):
    """out (B,S,H,D) running result, lse (B,S,H,1); block_lse (B,H,Sb).

    utils.py:25-26:  out -= sigmoid(blk_lse - lse) * (out - blk_out) ;
                     lse -= logsigmoid(lse - blk_lse)
    First call (out is None) just adopts the block (utils.py:38-42).  ``row_slice`` is the
    zigzag "rows c:" update (utils.py:43-48, call site zigzag_ring_flash_attn.py:61-67).
    """
    block_lse = np.swapaxes(block_lse, -2, -1)[..., None]   # (B,Sb,H,1)
    if out is None:
        assert row_slice is None, "first update_out_and_lse should not pass slice_ args"
        return np.array(block_out, dtype=np.float64), np.array(block_lse, dtype=np.float64)
    sl = (slice(None), row_slice if row_slice is not None else slice(None))
    o, l_ = out[sl], lse[sl]
    with np.errstate(invalid="ignore"):
        d = block_lse - l_
        d = np.where(np.isnan(d), -np.inf, d)             # (-inf) - (-inf): nothing to add
        new_o = o - _sigmoid(d) * (o - block_out)
        new_l = l_ - _logsigmoid(-d)
    out = out.copy()
    lse = lse.copy()
    out[sl], lse[sl] = new_o, new_l
    return out, lse


# ----------------------------------------------------------------------------
# Ulysses all-to-all, simulated over a list of per-rank arrays
# (yunchang/comm/all_to_all.py:15-102)
# ----------------------------------------------------------------------------
def all_to_all_4d_sim(xs: List[np.ndarray], scatter_idx: int, gather_idx: int) -> List[np.ndarray]:
    """xs[p] is ulysses-rank p's tensor.  scatter 2 / gather 1: (B,S/P,H,D) -> (B,S,H/P,D)
    (all_to_all.py:36-67); scatter 1 / gather 2: (B,S,H/P,D) -> (B,S/P,H,D) (:69-102).
    Rank p ends up with sequence chunks gathered in ulysses-rank order and head group p."""
    P = len(xs)
    if scatter_idx == 2 and gather_idx == 1:
        B, Sl, H, D = xs[0].shape
        assert H % P == 0
        hp = H // P
        return [np.concatenate([xs[src][:, :, p * hp:(p + 1) * hp] for src in range(P)], axis=1)
                for p in range(P)]
    if scatter_idx == 1 and gather_idx == 2:
        B, S, hp, D = xs[0].shape
        assert S % P == 0
        Sl = S // P
        return [np.concatenate([xs[src][:, p * Sl:(p + 1) * Sl] for src in range(P)], axis=2)
                for p in range(P)]
    raise RuntimeError("scatter_idx must be 1 or 2 and gather_idx must be 1 or 2")


# ----------------------------------------------------------------------------
# ring schedules, simulated over a list of per-ring-rank arrays
# ----------------------------------------------------------------------------
def zigzag_ring_forward_sim(qs, ks, vs, softmax_scale=None, dtype=np.float64):
    """zigzag_ring_flash_attn.py:5-76.  qs/ks/vs[r] are ring-rank r's (B,2c,H,D) tensors in
    zigzag layout [chunk r | chunk 2P-1-r].  Returns (outs, lses) with lse (B,H,2c)."""
    P = len(qs)
    outs, lses = [], []
    for r in range(P):
        q = qs[r]
        c = q.shape[1] // 2
        out = lse = None
        for step in range(P):
            src = (r - step) % P                       # K/V that arrived after `step` relays (utils.py:126-131)
            k, v = ks[src], vs[src]
            if step == 0:                               # :51-53
                bo, bl = block_fwd(q, k, v, softmax_scale, True, dtype)
                out, lse = update_out_and_lse(out, lse, bo, bl)
            elif step <= r:                             # :54-58
                bo, bl = block_fwd(q, k[:, :c], v[:, :c], softmax_scale, False, dtype)
                out, lse = update_out_and_lse(out, lse, bo, bl)
            else:                                       # :59-67
                bo, bl = block_fwd(q[:, c:], k, v, softmax_scale, False, dtype)
                out, lse = update_out_and_lse(out, lse, bo, bl, row_slice=slice(c, None))
        outs.append(out)
        lses.append(np.swapaxes(lse[..., 0], 1, 2))    # :74-76  (B,S,H,1) -> (B,H,S)
    return outs, lses


def zigzag_ring_backward_sim(douts, qs, ks, vs, outs, lses, softmax_scale=None, dtype=np.float64):
    """zigzag_ring_flash_attn.py:79-183 in exact arithmetic: every ring rank r computes block
    gradients against the K/V of every source rank; dq stays local, dk/dv are summed onto the
    owner (the travelling fp32 accumulators of :161-179 land on their owner after P steps)."""
    P = len(qs)
    dqs = [np.zeros(q.shape, dtype) for q in qs]
    dks = [np.zeros(k.shape, dtype) for k in ks]
    dvs = [np.zeros(v.shape, dtype) for v in vs]
    for r in range(P):
        q, do, o, lse = qs[r], douts[r], outs[r], lses[r]
        c = q.shape[1] // 2
        for step in range(P):
            src = (r - step) % P
            k, v = ks[src], vs[src]
            if step == 0:                               # :145-149
                dq, dk, dv = block_bwd(do, q, k, v, o, lse, softmax_scale, True, dtype)
                dqs[r] += dq; dks[src] += dk; dvs[src] += dv
            elif step <= r:                             # :151-155, :165-167
                dq, dk, dv = block_bwd(do, q, k[:, :c], v[:, :c], o, lse, softmax_scale, False, dtype)
                dqs[r] += dq; dks[src][:, :c] += dk; dvs[src][:, :c] += dv
            else:                                       # :156-159, :168-170
                dq, dk, dv = block_bwd(do[:, c:], q[:, c:], k, v, o[:, c:], lse[:, :, c:],
                                       softmax_scale, False, dtype)
                dqs[r][:, c:] += dq; dks[src] += dk; dvs[src] += dv
    return dqs, dks, dvs


def stripe_ring_forward_sim(qs, ks, vs, softmax_scale=None, dtype=np.float64):
    """stripe_flash_attn.py:6-76: every step is causal; steps > rank use q[:, 1:] x k[:, :-1] and
    update rows 1: only (:48-66)."""
    P = len(qs)
    outs, lses = [], []
    for r in range(P):
        q = qs[r]
        out = lse = None
        for step in range(P):
            src = (r - step) % P
            k, v = ks[src], vs[src]
            if step <= r:
                bo, bl = block_fwd(q, k, v, softmax_scale, True, dtype)
                out, lse = update_out_and_lse(out, lse, bo, bl)
            else:
                bo, bl = block_fwd(q[:, 1:], k[:, :-1], v[:, :-1], softmax_scale, True, dtype)
                out, lse = update_out_and_lse(out, lse, bo, bl, row_slice=slice(1, None))
        outs.append(out)
        lses.append(np.swapaxes(lse[..., 0], 1, 2))
    return outs, lses


def stripe_ring_backward_sim(douts, qs, ks, vs, outs, lses, softmax_scale=None, dtype=np.float64):
    """stripe_flash_attn.py:79-195 in exact arithmetic."""
    P = len(qs)
    dqs = [np.zeros(q.shape, dtype) for q in qs]
    dks = [np.zeros(k.shape, dtype) for k in ks]
    dvs = [np.zeros(v.shape, dtype) for v in vs]
    for r in range(P):
        q, do, o, lse = qs[r], douts[r], outs[r], lses[r]
        for step in range(P):
            src = (r - step) % P
            k, v = ks[src], vs[src]
            if step <= r:
                dq, dk, dv = block_bwd(do, q, k, v, o, lse, softmax_scale, True, dtype)
                dqs[r] += dq; dks[src] += dk; dvs[src] += dv
            else:
                dq, dk, dv = block_bwd(do[:, 1:], q[:, 1:], k[:, :-1], v[:, :-1], o[:, 1:], lse[:, :, 1:],
                                       softmax_scale, True, dtype)
                dqs[r][:, 1:] += dq; dks[src][:, :-1] += dk; dvs[src][:, :-1] += dv
    return dqs, dks, dvs


def basic_ring_forward_sim(qs, ks, vs, softmax_scale=None, causal=True, dtype=np.float64):
    """ring_flash_attn.py:7-62: contiguous layout; under causal only steps <= rank compute
    (:35) and only step 0 is causal (:41)."""
    P = len(qs)
    outs, lses = [], []
    for r in range(P):
        out = lse = None
        for step in range(P):
            src = (r - step) % P
            if (not causal) or step <= r:
                bo, bl = block_fwd(qs[r], ks[src], vs[src], softmax_scale, causal and step == 0, dtype)
                out, lse = update_out_and_lse(out, lse, bo, bl)
        outs.append(out)
        lses.append(np.swapaxes(lse[..., 0], 1, 2))
    return outs, lses


def basic_ring_backward_sim(douts, qs, ks, vs, outs, lses, softmax_scale=None, causal=True,
                            dtype=np.float64):
    """ring_flash_attn.py:65-149 in exact arithmetic."""
    P = len(qs)
    dqs = [np.zeros(q.shape, dtype) for q in qs]
    dks = [np.zeros(k.shape, dtype) for k in ks]
    dvs = [np.zeros(v.shape, dtype) for v in vs]
    for r in range(P):
        for step in range(P):
            src = (r - step) % P
            if (not causal) or step <= r:
                dq, dk, dv = block_bwd(douts[r], qs[r], ks[src], vs[src], outs[r], lses[r],
                                       softmax_scale, causal and step == 0, dtype)
                dqs[r] += dq; dks[src] += dk; dvs[src] += dv
    return dqs, dks, dvs


# ----------------------------------------------------------------------------
# packed variable-length batches  (yunchang/ring/zigzag_ring_flash_attn_varlen.py,
# yunchang/ring/ring_flash_attn_varlen.py, LSE layouts of yunchang/ring/utils.py:96-117)
#
# Token tensors are (T,H,D); sequence i owns rows cu_seqlens[i]:cu_seqlens[i+1].  Every step of the
# reference's varlen schedules acts on each sequence independently (the varlen flash kernels never mix
# sequences; get_half_index :27-42 and get_half_lse :45-58 cut every sequence in half exactly like the
# dense `c = S // 2`), so the oracle runs the dense schedule restatements above per sequence with B = 1.
# ----------------------------------------------------------------------------
def _cu(cu_seqlens) -> List[int]:
    return [int(c) for c in cu_seqlens]


def zigzag_extract_local_varlen(x: np.ndarray, cu_seqlens, rank: int, world_size: int) -> np.ndarray:
    """Ring rank `rank`'s shard of a packed (T,...) tensor: for every sequence, chunks `rank` and
    `2P-1-rank` of its 2P equal chunks (the per-sequence form of extract_local.py:29-49, ud = 1).
    Local cu_seqlens = cu_seqlens // P."""
    cu = _cu(cu_seqlens)
    parts = []
    for a, b in zip(cu[:-1], cu[1:]):
        assert (b - a) % (2 * world_size) == 0, "every sequence must split into 2*world_size chunks"
        ch = np.array_split(x[a:b], 2 * world_size, axis=0)
        parts += [ch[rank], ch[2 * world_size - 1 - rank]]
    return np.concatenate(parts, axis=0)


def basic_extract_local_varlen(x: np.ndarray, cu_seqlens, rank: int, world_size: int) -> np.ndarray:
    """Contiguous chunk `rank` of every sequence (per-sequence form of extract_local.py:25-26)."""
    cu = _cu(cu_seqlens)
    parts = []
    for a, b in zip(cu[:-1], cu[1:]):
        assert (b - a) % world_size == 0
        parts.append(np.array_split(x[a:b], world_size, axis=0)[rank])
    return np.concatenate(parts, axis=0)


def varlen_attention_ref(q, k, v, cu_seqlens, causal=True, softmax_scale=None, dtype=np.float64):
    """Per-sequence attention_ref over a packed batch.  Returns out (T,Hq,D), lse (Hq,T) -- the
    flattened LSE layout of utils.py:96-103."""
    cu = _cu(cu_seqlens)
    outs, lses = [], []
    for a, b in zip(cu[:-1], cu[1:]):
        o, l_ = attention_ref(q[None, a:b], k[None, a:b], v[None, a:b], causal, softmax_scale, dtype)
        outs.append(o[0]); lses.append(l_[0])
    return np.concatenate(outs, axis=0), np.concatenate(lses, axis=1)


def _per_sequence(fwd_sim, bwd_sim, cu_local, qs, ks, vs, douts, softmax_scale, dtype, **kw):
    cu = _cu(cu_local)
    P = len(qs)
    cut = lambda xs, a, b: [np.asarray(x[a:b], dtype)[None] for x in xs]
    res = {key: [[] for _ in range(P)] for key in ("out", "lse", "dq", "dk", "dv")}
    for a, b in zip(cu[:-1], cu[1:]):
        q_i, k_i, v_i = cut(qs, a, b), cut(ks, a, b), cut(vs, a, b)
        outs, lses = fwd_sim(q_i, k_i, v_i, softmax_scale, dtype=dtype, **kw)
        for r in range(P):
            res["out"][r].append(outs[r][0]); res["lse"][r].append(lses[r][0])
        if douts is not None:
            dqs, dks, dvs = bwd_sim(cut(douts, a, b), q_i, k_i, v_i, outs, lses, softmax_scale,
                                    dtype=dtype, **kw)
            for r in range(P):
                res["dq"][r].append(dqs[r][0]); res["dk"][r].append(dks[r][0]); res["dv"][r].append(dvs[r][0])
    cat = lambda key, ax: [np.concatenate(x, axis=ax) for x in res[key]] if res[key][0] else None
    return cat("out", 0), cat("lse", 1), cat("dq", 0), cat("dk", 0), cat("dv", 0)


def zigzag_ring_varlen_sim(qs, ks, vs, cu_seqlens_local, douts=None, softmax_scale=None, dtype=np.float64):
    """zigzag_ring_flash_attn_varlen.py:61-157 (forward) and :160-290 (backward).  qs/ks/vs[r]: ring
    rank r's local packed (T_local,H,D) tensors, every sequence in zigzag layout.  Returns per-rank
    (outs (T,H,D), lses (H,T), dqs, dks, dvs); gradients are None without `douts`."""
    return _per_sequence(zigzag_ring_forward_sim, zigzag_ring_backward_sim, cu_seqlens_local, qs, ks, vs,
                         douts, softmax_scale, dtype)


def basic_ring_varlen_sim(qs, ks, vs, cu_seqlens_local, douts=None, softmax_scale=None, causal=True,
                          dtype=np.float64):
    """ring_flash_attn_varlen.py:28-85 (forward) and :88-176 (backward): contiguous shards."""
    return _per_sequence(basic_ring_forward_sim, basic_ring_backward_sim, cu_seqlens_local, qs, ks, vs,
                         douts, softmax_scale, dtype, causal=causal)


# ----------------------------------------------------------------------------
# the whole hybrid layer, simulated  (yunchang/hybrid/attn_layer.py:57-161)
# ----------------------------------------------------------------------------
def usp_forward_sim(local_qs, local_ks, local_vs, ud, rd, ring_impl_type="zigzag", causal=True,
                    softmax_scale=None, dtype=np.float64, return_ctx=False):
    """local_*[rank] are the per-rank shards as EXTRACT_FUNC_DICT produces them.  Returns the
    per-rank outputs (B,S/ws,Hq,D) of LongContextAttention.forward:
    3x a2a (attn_layer.py:111-119) -> ring fn (:132-147) -> a2a back (:156-158)."""
    ws = len(local_qs)
    ulysses, ring = seq_parallel_groups(ud, rd, ws)
    hq, hk, hv = [None] * ws, [None] * ws, [None] * ws
    for grp in ulysses:
        for name, src, dst in (("q", local_qs, hq), ("k", local_ks, hk), ("v", local_vs, hv)):
            res = all_to_all_4d_sim([np.asarray(src[r], dtype) for r in grp], 2, 1)
            for i, r in enumerate(grp):
                dst[r] = res[i]
    ring_out, ring_lse = [None] * ws, [None] * ws
    for grp in ring:
        qs, ks, vs = [hq[r] for r in grp], [hk[r] for r in grp], [hv[r] for r in grp]
        if ring_impl_type == "zigzag":
            assert causal, "zigzag ring is meaningless for causal=False"
            o, l_ = zigzag_ring_forward_sim(qs, ks, vs, softmax_scale, dtype)
        elif ring_impl_type == "strip":
            assert causal, "stripe flash attn only supports causal attention"
            o, l_ = stripe_ring_forward_sim(qs, ks, vs, softmax_scale, dtype)
        else:
            o, l_ = basic_ring_forward_sim(qs, ks, vs, softmax_scale, causal, dtype)
        for i, r in enumerate(grp):
            ring_out[r], ring_lse[r] = o[i], l_[i]
    outs = [None] * ws
    for grp in ulysses:
        res = all_to_all_4d_sim([ring_out[r] for r in grp], 1, 2)
        for i, r in enumerate(grp):
            outs[r] = res[i]
    if return_ctx:
        return outs, dict(hq=hq, hk=hk, hv=hv, ring_out=ring_out, ring_lse=ring_lse,
                          ulysses=ulysses, ring=ring)
    return outs


def usp_backward_sim(local_douts, ctx, ud, rd, ring_impl_type="zigzag", causal=True,
                     softmax_scale=None, dtype=np.float64):
    """Backward of the hybrid layer: a2a^-1 of dout (all_to_all.py:124-134), ring backward,
    then a2a^-1 of dq/dk/dv."""
    ws = len(local_douts)
    ulysses, ring = ctx["ulysses"], ctx["ring"]
    hdo = [None] * ws
    for grp in ulysses:
        res = all_to_all_4d_sim([np.asarray(local_douts[r], dtype) for r in grp], 2, 1)
        for i, r in enumerate(grp):
            hdo[r] = res[i]
    hdq, hdk, hdv = [None] * ws, [None] * ws, [None] * ws
    for grp in ring:
        args = ([hdo[r] for r in grp], [ctx["hq"][r] for r in grp], [ctx["hk"][r] for r in grp],
                [ctx["hv"][r] for r in grp], [ctx["ring_out"][r] for r in grp],
                [ctx["ring_lse"][r] for r in grp])
        if ring_impl_type == "zigzag":
            dq, dk, dv = zigzag_ring_backward_sim(*args, softmax_scale, dtype)
        elif ring_impl_type == "strip":
            dq, dk, dv = stripe_ring_backward_sim(*args, softmax_scale, dtype)
        else:
            dq, dk, dv = basic_ring_backward_sim(*args, softmax_scale, causal, dtype)
        for i, r in enumerate(grp):
            hdq[r], hdk[r], hdv[r] = dq[i], dk[i], dv[i]
    dqs, dks, dvs = [None] * ws, [None] * ws, [None] * ws
    for grp in ulysses:
        for src, dst in ((hdq, dqs), (hdk, dks), (hdv, dvs)):
            res = all_to_all_4d_sim([src[r] for r in grp], 1, 2)
            for i, r in enumerate(grp):
                dst[r] = res[i]
    return dqs, dks, dvs


# ----------------------------------------------------------------------------
# dtype helpers (bf16 has no numpy dtype: carried as uint16 bit patterns)
# ----------------------------------------------------------------------------
def bf16_bits_to_f32(u16: np.ndarray) -> np.ndarray:
    return (u16.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even, as torch's .to(bfloat16)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    rounding = ((u >> 16) & 1) + np.uint32(0x7FFF)
    out = ((u + rounding) >> 16).astype(np.uint16)
    nan = np.isnan(x)
    out[nan] = 0x7FC0
    return out


def round_bf16(x: np.ndarray) -> np.ndarray:
    return bf16_bits_to_f32(f32_to_bf16_bits(x)).astype(np.float64)
