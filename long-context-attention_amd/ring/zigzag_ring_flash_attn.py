"""Zigzag ring attention: same surface as yunchang/ring/zigzag_ring_flash_attn.py.

Schedule (identical block structure to the reference, :45-72 forward / :139-179 backward): with
the local sequence = [chunk r | chunk 2P-1-r] of length 2c, ring step `s` sees the K/V of ring rank
r-s and computes
    s == 0 : causal  q[0:2c] x k[0:2c]
    s <= r : full    q[0:2c] x k[0:c]
    s >  r : full    q[c:2c] x k[0:2c]          (only rows c: are updated)
so every step costs 2c^2 score entries on every rank.

MI355X-first differences (results identical up to fp32 rounding order):
  * the LSE merge (`update_out_and_lse`, ring/utils.py:10-51) is fused into the attention kernel's
    epilogue: the block result never makes a 16-bit round trip through HBM, the running output is
    fp32 (`acc`) and the LSE stays fp32 (the reference TORCH_* path rounds it to bf16,
    kernels/attention.py:135); rows are emitted in 16-bit exactly once, by the step that
    finalises them (rows [0,c) at step r, rows [c,2c) at step P-1);
  * the backward kernels accumulate dQ in place in fp32 and emit fp32 dK/dV blocks, replacing the
    16-bit dq/dk/dv buffers + fp32 adds of :115-170;
  * K/V are relayed ahead of the kernels on a side HIP stream (ring/utils.py KVRelay).
"""
import torch
import torch.distributed as dist

from ..kernels import AttnType
from ..kernels.attention import get_block_backend, kernel_head_dim, kernel_operand, needs_grad, pad_head_dim
from .utils import FULL, KVRelay, group_info, ZigzagKVFetch, final_grads, kv_relay_mode, travel_dkdv, zigzag_fetch_pieces



def zigzag_fwd_step(be, r, P, step, q, kk, vv, softmax_scale, lse, out, acc):
    """One ring step of the forward on ring rank `r` of `P`, given the K/V that arrived after
    `step` hops.  Pure schedule logic (no communication): also driven by the single-GPU tests,
    which emulate the ring with virtual ranks."""
    S2 = q.shape[1]
    c = S2 // 2
    last = step == P - 1
    if step == 0:                                   # zigzag_ring_flash_attn.py:51-53
        fe = S2 if last else (c if r == 0 else 0)
        be.fwd(q, kk, vv, softmax_scale, True, lse, out, acc, False, 0, fe)
    elif step <= r:                                 # :54-58
        fe = S2 if last else (c if step == r else 0)
        be.fwd(q, kk[:, :c], vv[:, :c], softmax_scale, False, lse, out, acc, True, 0, fe)
    else:                                           # :59-67
        be.fwd(q[:, c:], kk, vv, softmax_scale, False, lse[:, :, c:], out[:, c:], acc[:, c:], True,
               0, c if last else 0)


def zigzag_fwd_step0_own(be, r, u, own, softmax_scale, lse, out, acc):
    """Step 0 of the forward started on the rows this rank already holds (hybrid/async_attn_layer.py: self_chunk_mode;
    ulysses degree 2 beside a ring).  The local block is [front chunk | back chunk] of c rows each and the Ulysses
    exchange delivers one of them from the peer: rank u = 0 owns the front chunk, u = 1 the back chunk.  Step 0 is
    causal over the 2c local rows (zigzag_ring_flash_attn.py:51-53), so the owned chunk against its own keys is a
    complete sub-block of it: u = 0: q[0:c] x k[0:c] causal = everything rows [0,c) get from step 0; u = 1: the
    diagonal block of rows [c,2c).  `own` = (q, k, v) of the owned chunk (views of the exchange's send buffer)."""
    qs, ks, vs = own
    c = qs.shape[1]
    if u == 0:           # rows [0,c) are final at once on ring rank 0 only (the first chunk of the whole sequence)
        be.fwd(qs, ks, vs, softmax_scale, True, lse[:, :, :c], out[:, :c], acc[:, :c], False, 0, c if r == 0 else 0)
    else:
        be.fwd(qs, ks, vs, softmax_scale, True, lse[:, :, c:], out[:, c:], acc[:, c:], False, 0, 0)


def zigzag_fwd_step0_rest(be, r, u, q, k, v, softmax_scale, lse, out, acc):
    """... and the rest of step 0 behind the exchange: u = 0: rows [c,2c) against all 2c local keys (bottom-right
    causal); u = 1: the peer's keys [0,c) merged into rows [c,2c) (full block), then rows [0,c) causal."""
    c = q.shape[1] // 2
    if u == 0:
        be.fwd(q[:, c:], k, v, softmax_scale, True, lse[:, :, c:], out[:, c:], acc[:, c:], False, 0, 0)
    else:
        be.fwd(q[:, c:], k[:, :c], v[:, :c], softmax_scale, False, lse[:, :, c:], out[:, c:], acc[:, c:], True, 0, 0)
        be.fwd(q[:, :c], k[:, :c], v[:, :c], softmax_scale, True, lse[:, :, :c], out[:, :c], acc[:, :c], False, 0,
               c if r == 0 else 0)


def zigzag_bwd_step0_split(be, u, do_own, dout, wait, q, kk, vv, out, lse, delta, softmax_scale, dq_acc, dk_dst, dv_dst):
    """Step 0 of the backward started on the rows whose dO this rank already holds (the owned chunk; K, V, out and the
    LSE are saved tensors): u = 0: rows [0,c) x keys [0,c) first, rows [c,2c) x all keys on top behind the exchange;
    u = 1: rows [c,2c) x all keys first (3/4 of the block), rows [0,c) x keys [0,c) on top.  dk_dst / dv_dst are the
    travelling fp32 accumulators (written by the first launch, accumulated into by the second); the delta of the peer's
    rows is computed behind the exchange as well."""
    c = q.shape[1] // 2
    if u == 0:
        be.delta(do_own, out[:, :c], delta[:, :, :c])
        dk_dst[:, c:].zero_()
        dv_dst[:, c:].zero_()
        be.bwd(do_own, q[:, :c], kk[:, :c], vv[:, :c], lse[:, :, :c], delta[:, :, :c], dq_acc[:, :c], dk_dst[:, :c], dv_dst[:, :c],
               softmax_scale, True)
        wait()
        be.delta(dout[:, c:], out[:, c:], delta[:, :, c:])
        be.bwd(dout[:, c:], q[:, c:], kk, vv, lse[:, :, c:], delta[:, :, c:], dq_acc[:, c:], dk_dst, dv_dst, softmax_scale, True,
               accum_dk=True, accum_dv=True)
    else:
        be.delta(do_own, out[:, c:], delta[:, :, c:])
        be.bwd(do_own, q[:, c:], kk, vv, lse[:, :, c:], delta[:, :, c:], dq_acc[:, c:], dk_dst, dv_dst, softmax_scale, True)
        wait()
        be.delta(dout[:, :c], out[:, :c], delta[:, :, :c])
        be.bwd(dout[:, :c], q[:, :c], kk[:, :c], vv[:, :c], lse[:, :, :c], delta[:, :, :c], dq_acc[:, :c], dk_dst[:, :c],
               dv_dst[:, :c], softmax_scale, True, accum_dk=True, accum_dv=True)


def zigzag_bwd_block(be, r, P, step, dout, q, kk, vv, lse, delta, softmax_scale, dq_acc, dk_dst,
                     dv_dst, only=None, dq16=None):
    """Block backward of ring step `step`: dq accumulates in place into dq_acc (fp32), the dK/dV
    block is written to dk_dst/dv_dst (fp32; at step 0 these ARE the travelling accumulators).
    `only` = "dq" | "dkdv": just that launch of the two (steps > 0).  `dq16` (with only="dq"; the LAST step): dq is final
    behind this launch -- accumulator + block, rounded in the epilogue into dq16's rows; the rows this step does not touch
    are cast from the accumulator."""
    c = q.shape[1] // 2
    kw = {} if only is None else {"only": only}
    nz = lambda t, sl: None if t is None else t[:, sl]
    if step == 0:                                   # zigzag_ring_flash_attn.py:145-149
        be.bwd(dout, q, kk, vv, lse, delta, dq_acc, dk_dst, dv_dst, softmax_scale, True, **kw)
    elif step <= r:                                 # :151-155
        be.bwd(dout, q, kk[:, :c], vv[:, :c], lse, delta, dq_acc, nz(dk_dst, slice(0, c)), nz(dv_dst, slice(0, c)),
               softmax_scale, False, accum_dq=True, dq16=dq16, **kw)
    else:                                           # :156-159
        if dq16 is not None:
            be.cast(dq16[:, :c], dq_acc[:, :c])     # final since step r
        be.bwd(dout[:, c:], q[:, c:], kk, vv, lse[:, :, c:], delta[:, :, c:], nz(dq_acc, slice(c, None)), dk_dst,
               dv_dst, softmax_scale, False, accum_dq=True, dq16=nz(dq16, slice(c, None)), **kw)


def split_steps() -> bool:
    """USP_BWD_SPLIT_STEPS=0: one dK/dV + dQ call per ring step as in rounds 1-5 (A/B switch; the default issues the dQ launch of
    a step behind the posting of the step's hop)."""
    import os
    return os.environ.get("USP_BWD_SPLIT_STEPS", "1") != "0"


def zigzag_bwd_fold(be, r, step, c, dk_acc, dv_acc, dk_blk, dv_blk):
    """Add this step's dK/dV block into the accumulators that just arrived (:161-170)."""
    if step <= r:
        be.add(dk_acc[:, :c], dk_acc[:, :c], dk_blk[:, :c])
        be.add(dv_acc[:, :c], dv_acc[:, :c], dv_blk[:, :c])
    else:
        be.add(dk_acc, dk_acc, dk_blk)
        be.add(dv_acc, dv_acc, dv_blk)


def zigzag_fetch_plan(P, r, pieces, c, grouped=False):
    """The launches of ring rank r behind step 0 when the K/V arrive through ZigzagKVFetch: (wave, first step, last step,
    all_rows, final_end) in issue order; a launch reads the wave's piece of the ring ranks r - first ... r - last.
    WAVE-major: whatever needs only the waves that have landed runs before the compute stream waits for the next one
    (step-major, rank 0 would sit behind the last wave from its first step on while the front-half launches of its other
    steps were ready).  Steps s <= r read the front waves with every q row (zigzag_ring_flash_attn.py:54-58), steps s > r
    read all waves with q[c:] (:59-67; final_end then counts rows of q[c:]).  `grouped` (batch 1: the pieces of a wave
    lie one behind the other): the steps of a wave that share a query range are ONE launch -- at most two per wave
    instead of P - 1, each with one merge epilogue.  Rows are emitted in 16 bits by the LAST launch that touches them:
    rows [0,c) by the last all-rows launch, rows [c,2c) by the last launch of all (pure schedule logic;
    tests/test_host_api.py)."""
    launches = []
    for w in range(2 * pieces):
        if grouped:
            if w < pieces and r >= 1:
                launches.append((w, 1, r, True))
            if r < P - 1:
                launches.append((w, r + 1, P - 1, False))
        else:
            launches += [(w, step, step, step <= r) for step in range(1, P) if w < pieces or step > r]
    last_any = len(launches) - 1
    last_all_rows = max((i for i, l in enumerate(launches) if l[3]), default=-1)
    plan = []
    for i, (w, lo, hi, all_rows) in enumerate(launches):
        if all_rows:
            fe = 2 * c if i == last_any else (c if i == last_all_rows else 0)
        else:
            fe = c if i == last_any else 0
        plan.append((w, lo, hi, all_rows, fe))
    return plan


def tail_k_splits(B, H, rows, keys):
    """K cuts of a row-chunked tail launch (forward; non-causal, merge mode): a chunk of `rows` query rows has B * H *
    ceil(rows / 256) work items, every one as long as the others -- below one per CU the idle CUs are filled by cutting every
    item's keys (usp_fwd_args.k_splits; cuts of at least 1024 keys, the merge launch costs a few microseconds)."""
    from ..comm.link import device_cus
    items = B * H * ((rows + 255) // 256)
    n = min(8, device_cus() // max(1, items), keys // 1024)
    return n if n > 1 else 0


def _final_rows(be, q, kp, vp, softmax_scale, lse, out, acc, lo, hi, tail):
    """The launch that FINALISES rows [lo, hi) of the block (a full, merge-mode launch against kp / vp; [lo, hi) is the whole
    block or its back half: all of them emitted in 16 bits) as `tail` = (n, emit) asks for it: n launches over row pieces
    -- piece j of every c-row chunk in [lo, hi) -- with `emit(j, out)` called behind the launches of piece j: from then on
    rows [chunk * c + j c / n, chunk * c + (j + 1) c / n) of EVERY chunk of the block are final on the calling stream, and the
    caller's output exchange of those rows runs beside the launches of the pieces that follow (hybrid/async_attn_layer.py:
    row-chunked tails; the last-out exchange of a pass is the one no head-group pipeline can hide)."""
    n, emit = tail
    c = q.shape[1] // 2
    B, _, H, _ = q.shape
    for j in range(n):
        for ch in range(lo // c, hi // c):
            a, b = ch * c + j * c // n, ch * c + (j + 1) * c // n
            if b > a:
                be.fwd(q[:, a:b], kp, vp, softmax_scale, False, lse[:, :, a:b], out[:, a:b], acc[:, a:b], True, 0, b - a,
                       k_splits=tail_k_splits(B, H, b - a, kp.shape[1]))
        emit(j, out)


def zigzag_forward_phases(process_group, q, k, v, softmax_scale, overlap=False, first=None, tail=None):
    """The zigzag ring forward as a generator of two phases.  With `first` it yields ONCE, behind the launch on the owned chunk
    and in front of the wait for the caller's exchange -- the caller may start other head groups' owned chunks there -- and
    returns (out, lse) through StopIteration; without `first` it never yields.  zigzag_ring_flash_attn_forward drives it to the end."""
    P, r = group_info(dist, process_group)
    be = get_block_backend(beside_transfers=P > 1 or overlap)
    B, S2, H, D = q.shape
    assert S2 % 2 == 0, "zigzag layout needs an even local sequence length"
    dev = q.device
    out = torch.empty((B, S2, H, D), dtype=q.dtype, device=dev)
    lse = torch.empty((B, H, S2), dtype=torch.float32, device=dev)
    if P == 1:           # one block, no relay, no running accumulator
        be.fwd(q, k, v, softmax_scale, True, lse, out)
        return out, lse
    acc = torch.empty((B, S2, H, D), dtype=torch.float32, device=dev)
    c = S2 // 2
    if first is not None:
        u, own, wait = first
        zigzag_fwd_step0_own(be, r, u, own, softmax_scale, lse, out, acc)
        yield
        wait()

    def step0():
        if first is not None:
            zigzag_fwd_step0_rest(be, r, first[0], q, k, v, softmax_scale, lse, out, acc)
        else:
            zigzag_fwd_step(be, r, P, 0, q, k, v, softmax_scale, lse, out, acc)

    if P > 2 and kv_relay_mode(P) == "direct":      # mesh fetch in waves, only the halves the schedule reads
        with ZigzagKVFetch(process_group, k, v, zigzag_fetch_pieces(k)) as fetch:
            step0()
            plan = zigzag_fetch_plan(P, r, fetch.pieces, c, fetch.grouped)
            for i, (w, lo, hi, all_rows, fe) in enumerate(plan):
                kp, vp = fetch.get_range(w, lo, hi) if fetch.grouped else fetch.get(w, lo)
                if tail is not None and i == len(plan) - 1:       # the last launch of all finalises every row it computes
                    _final_rows(be, q, kp, vp, softmax_scale, lse, out, acc, 0 if all_rows else c, S2, tail)
                elif all_rows:
                    be.fwd(q, kp, vp, softmax_scale, False, lse, out, acc, True, 0, fe)
                else:
                    be.fwd(q[:, c:], kp, vp, softmax_scale, False, lse[:, :, c:], out[:, c:], acc[:, c:], True, 0, fe)
        return out, lse
    with KVRelay(process_group, k, v) as relay:
        for step in range(P):
            kk, vv = relay.get(step)
            if step == 0:
                step0()
            elif tail is not None and step == P - 1:              # (step <= r: every row x the front keys; else the back rows)
                if step <= r:
                    _final_rows(be, q, kk[:, :c], vv[:, :c], softmax_scale, lse, out, acc, 0, S2, tail)
                else:
                    _final_rows(be, q, kk, vv, softmax_scale, lse, out, acc, c, S2, tail)
            else:
                zigzag_fwd_step(be, r, P, step, q, kk, vv, softmax_scale, lse, out, acc)
    return out, lse


def zigzag_ring_flash_attn_forward(process_group, q, k, v, softmax_scale, dropout_p=0, causal=True,
                                   window_size=(-1, -1), softcap=0.0, alibi_slopes=None,
                                   deterministic=False, attn_type: AttnType = AttnType.HIP, overlap=False, first=None, tail=None):
    """`overlap`: the caller has transfers of its own in flight (pipelined Ulysses exchange), so the kernels are
    launched so that collectives can run beside them even at ring degree 1.
    `first` = (u, (q, k, v) of the owned chunk, wait): q / k / v are still in flight (the caller's Ulysses exchange at degree
    2; `wait()` orders the calling stream behind it) and step 0 starts on the chunk this rank owns (zigzag_fwd_step0_own).
    The K/V transfers of the ring are posted BEHIND the wait -- they read the exchanged k / v -- i.e. no later than without
    the split, where everything waits for the exchange.  Ring degree > 1 only (degree 1: _split_first_forward).
    `tail` = (n, emit): the launch that finalises the last rows runs in n row pieces, `emit(j, out)` behind piece j (_final_rows);
    ring degree > 1 only."""
    assert causal == True, "zigzag ring is meaningless for causal=False"
    gen = zigzag_forward_phases(process_group, q, k, v, softmax_scale, overlap, first, tail)
    try:
        while True:
            next(gen)
    except StopIteration as done:
        return done.value


def zigzag_ring_flash_attn_backward(process_group, dout, q, k, v, out, softmax_lse, softmax_scale,
                                    dropout_p=0, causal=True, window_size=(-1, -1), softcap=0.0,
                                    alibi_slopes=None, deterministic=False,
                                    attn_type: AttnType = AttnType.HIP, overlap=False, tail=None, first=None, dq_first=None):
    """`dq_first(dq)`: the LAST ring step issues its dQ launch in front of its dK/dV launch and rounds dQ in that launch's
    epilogue (rows the step does not touch are cast from the accumulator); `dq_first` is called with the final 16-bit dq
    between the two launches, so that the caller's exchange of dq -- most of the bytes of the gradient exchange -- runs beside
    the dK/dV launch instead of behind the whole step.  Ring degree > 1 only.
    `first` = (u, dO of the owned chunk, wait): dO is still in flight (see the forward); step 0 starts on the owned
    rows (zigzag_bwd_step0_split).  Ring degree > 1 only.  Ordering differs from the forward's: k and v are SAVED tensors
    here, so travel_dkdv enters the K/V relay -- which posts the ring's transfers -- BEFORE block(0) calls `wait()` on the dO
    exchange; the ring's and the Ulysses communicator therefore have transfers in flight together from step 0 on (in the forward
    the ring's transfers read exchanged tensors and are posted behind the wait).  Correct either way -- nothing the relay moves
    depends on dO -- and exercised through RCCL on one-device virtual grids only (tests/test_gpu_rccl_order.py); a first run on
    real devices that stalls here should set USP_SELF_CHUNK=0."""
    assert causal == True, "zigzag ring is meaningless for causal=False"
    P, r = group_info(dist, process_group)
    be = get_block_backend(beside_transfers=P > 1 or overlap)
    B, S2, H, D = q.shape
    c = S2 // 2
    dev = q.device
    lse = softmax_lse
    delta = torch.empty((B, H, S2), dtype=torch.float32, device=dev)
    assert first is None or P > 1
    if first is None:
        be.delta(dout, out, delta)
    if P == 1:   # one block: the kernels round the gradients to q.dtype in their epilogues
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        be.bwd(dout, q, k, v, lse, delta, None, None, None, softmax_scale, True, dq16=dq, dk16=dk, dv16=dv)
        return dq, dk, dv
    dq_acc = torch.empty((B, S2, H, D), dtype=torch.float32, device=dev)

    dq_done = []

    # Every step is issued as dK/dV launch | fold + hop posted | dQ launch (travel_dkdv: split_block), so a step's hop also runs
    # beside its own dQ launch -- the LAST hop included, which on a ring-only grid nothing else hides.  Two exceptions: a step 0
    # that starts on the owned rows (`first`) keeps its two launch pairs together, and the last step under `dq_first` runs dQ
    # FIRST (its dq exchange is the bigger transfer; the hop then follows the dK/dV launch as before).
    def block(step, kk, vv, dk_dst, dv_dst, only=None):
        if step == 0 and first is not None:
            if only == "dq":
                return
            u, do_own, wait = first
            zigzag_bwd_step0_split(be, u, do_own, dout, wait, q, kk, vv, out, lse, delta, softmax_scale, dq_acc, dk_dst, dv_dst)
        elif dq_first is not None and step == P - 1:
            if only == "dq":
                return
            dq16 = torch.empty((B, S2, H, D), dtype=q.dtype, device=dev)
            zigzag_bwd_block(be, r, P, step, dout, q, kk, vv, lse, delta, softmax_scale, dq_acc, None, None, only="dq", dq16=dq16)
            dq_first(dq16)
            dq_done.append(dq16)
            zigzag_bwd_block(be, r, P, step, dout, q, kk, vv, lse, delta, softmax_scale, None, dk_dst, dv_dst, only="dkdv")
        elif only == "dkdv":
            zigzag_bwd_block(be, r, P, step, dout, q, kk, vv, lse, delta, softmax_scale, None, dk_dst, dv_dst, only="dkdv")
        elif only == "dq":
            zigzag_bwd_block(be, r, P, step, dout, q, kk, vv, lse, delta, softmax_scale, dq_acc, None, None, only="dq")
        else:
            zigzag_bwd_block(be, r, P, step, dout, q, kk, vv, lse, delta, softmax_scale, dq_acc, dk_dst, dv_dst)

    def fold(step, dk_acc, dv_acc, dk_blk, dv_blk):
        zigzag_bwd_fold(be, r, step, c, dk_acc, dv_acc, dk_blk, dv_blk)

    # steps s <= rank carry gradients for the front-half K/V rows only (:151-155, :161-170)
    dk_acc, dv_acc = travel_dkdv(process_group, k, v, block, fold, be=be, final_dtype=k.dtype, defer=tail,
                                 extent=lambda rank, step: slice(0, c) if step <= rank else FULL, split_block=split_steps())
    return final_grads(be, (q, k, v), (dq_done[0] if dq_done else dq_acc, dk_acc, dv_acc))


class ZigZagRingFlashAttnFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes,
                deterministic, return_softmax, group, attn_type):
        if softmax_scale is None:
            softmax_scale = q.shape[-1] ** (-0.5)
        assert alibi_slopes is None
        q, k, v = kernel_operand(q), kernel_operand(k), kernel_operand(v)     # any view a caller holds (maybe_contiguous)
        _check_hot_path_args(dropout_p, window_size, softcap)
        out, softmax_lse = zigzag_ring_flash_attn_forward(
            group, q, k, v, softmax_scale=softmax_scale, dropout_p=dropout_p, causal=causal,
            window_size=window_size, softcap=softcap, alibi_slopes=alibi_slopes, deterministic=False,
            attn_type=attn_type)
        ctx.save_for_backward(q, k, v, out, softmax_lse)
        ctx.dropout_p = dropout_p
        ctx.softmax_scale = softmax_scale
        ctx.causal = causal
        ctx.window_size = window_size
        ctx.softcap = softcap
        ctx.alibi_slopes = alibi_slopes
        ctx.deterministic = deterministic
        ctx.group = group
        ctx.attn_type = attn_type
        return out if not return_softmax else (out, softmax_lse, None)

    @staticmethod
    def backward(ctx, dout, *args):
        dout = kernel_operand(dout)
        q, k, v, out, softmax_lse = ctx.saved_tensors
        dq, dk, dv = zigzag_ring_flash_attn_backward(
            ctx.group, dout, q, k, v, out, softmax_lse, softmax_scale=ctx.softmax_scale,
            dropout_p=ctx.dropout_p, causal=ctx.causal, window_size=ctx.window_size,
            softcap=ctx.softcap, alibi_slopes=ctx.alibi_slopes, deterministic=ctx.deterministic,
            attn_type=ctx.attn_type)
        return dq, dk, dv, None, None, None, None, None, None, None, None, None, None


def _check_hot_path_args(dropout_p, window_size, softcap):
    if dropout_p not in (0, 0.0):
        raise NotImplementedError("dropout_p != 0 is not supported by the HIP ring attention")
    if window_size is not None and tuple(window_size) != (-1, -1):
        raise NotImplementedError("sliding-window attention is not supported by the HIP ring attention")
    if softcap not in (None, 0, 0.0):
        raise NotImplementedError("softcap is not supported by the HIP ring attention")


def zigzag_ring_flash_attn_qkvpacked_func(qkv, dropout_p=0.0, softmax_scale=None, causal=False,
                                          window_size=(-1, -1), softcap=0.0, alibi_slopes=None,
                                          deterministic=False, return_attn_probs=False, group=None,
                                          attn_type: AttnType = AttnType.HIP):
    return ZigZagRingFlashAttnFunc.apply(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], dropout_p,
                                         softmax_scale, causal, window_size, softcap, alibi_slopes,
                                         deterministic, return_attn_probs, group, attn_type)


def zigzag_ring_flash_attn_kvpacked_func(q, kv, dropout_p=0.0, softmax_scale=None, causal=False,
                                         window_size=(-1, -1), softcap=0.0, alibi_slopes=None,
                                         deterministic=False, return_attn_probs=False, group=None,
                                         attn_type: AttnType = AttnType.HIP):
    return ZigZagRingFlashAttnFunc.apply(q, kv[:, :, 0], kv[:, :, 1], dropout_p, softmax_scale,
                                         causal, window_size, softcap, alibi_slopes, deterministic,
                                         return_attn_probs, group, attn_type)


def zigzag_ring_flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False,
                                window_size=(-1, -1), softcap=0.0, alibi_slopes=None,
                                deterministic=False, return_attn_probs=False, group=None,
                                attn_type: AttnType = AttnType.HIP, attn_processor=None):
    D = q.shape[-1]
    if kernel_head_dim(D) != D:      # a head dim the kernels do not instantiate (e.g. 96): zero-padded copies
        res = zigzag_ring_flash_attn_func(*pad_head_dim(q, k, v), dropout_p, D ** -0.5 if softmax_scale is None else softmax_scale, causal,
                                          window_size, softcap, alibi_slopes, deterministic, return_attn_probs, group, attn_type, attn_processor)
        return (res[0][..., :D],) + tuple(res[1:]) if isinstance(res, tuple) else res[..., :D]
    if not needs_grad(q, k, v):      # inference / forward-only benchmarks: no autograd node, no saved tensors
        assert alibi_slopes is None
        _check_hot_path_args(dropout_p, window_size, softcap)
        out, lse = zigzag_ring_flash_attn_forward(
            group, kernel_operand(q), kernel_operand(k), kernel_operand(v),
            softmax_scale=q.shape[-1] ** (-0.5) if softmax_scale is None else softmax_scale, causal=causal,
            attn_type=attn_type)
        return out if not return_attn_probs else (out, lse, None)
    return ZigZagRingFlashAttnFunc.apply(q, k, v, dropout_p, softmax_scale, causal, window_size,
                                         softcap, alibi_slopes, deterministic, return_attn_probs,
                                         group, attn_type)
