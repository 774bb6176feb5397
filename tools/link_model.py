"""Link arithmetic for the multi-GPU BASELINE configs (DEV TOOL, no GPU needed).

What one rank's iteration should cost on a real node, from (a) the per-rank compute-only times measured on one MI355X
(tools/rank_emulation.py, profiles/r02_rank_emulation.txt), (b) the bytes each schedule puts on a link and (c) an
assumed xGMI rate per link and direction.  It replays the ORDER the package issues things in -- every input exchange of
the head groups first on the "ulysses" lane, group i's output exchange behind its attention; the ring K/V fetch in one
or two waves over P-1 links in parallel; the travelling dK/dV one hop per step -- with two resources, the lane(s) and
the compute stream.  Nothing here is a measurement; it is the expectation the driver's N = 2/4/8 runs can be held
against."""
import argparse

MiB = 2 ** 20


def pipeline(t_in, t_comp, t_out, ng, eff=1.0):
    """ng head groups: inputs queued up-front on the lane, outputs as their attention finishes.  Returns (total,
    exposed = total - compute)."""
    i, c, o = t_in / ng, t_comp / eff / ng, t_out / ng
    lane = comp = 0.0
    landed = []
    for _ in range(ng):
        lane += i
        landed.append(lane)
    for g in range(ng):
        comp = max(comp, landed[g]) + c
        lane = max(lane, comp) + o
    return lane, lane - t_comp / eff


def round5(ms):
    """Round 5: what `bench.py --gpus N` runs -- the metric's configuration (B1 S65536 H32/Hkv4 D128 bf16 causal fwd+bwd, 123.15
    TFLOP per iteration) on the grids 1x1, 2x1, 1x4, 2x4 -- from the compute-only iteration of one rank measured on one MI355X
    with round 5's kernels (profiles/r05_rank_emulation.txt; N = 1: the driver's own line, profiles/r05_bench_driver_cmd_*.json)
    and the bytes each schedule puts on a link.  fwd : bwd = 0.235 : 0.765 (28.4 : 92.7 ms on one GPU)."""
    F = 123.15
    print("\nround 5 bench workloads (B1 S65536 H32/Hkv4 fwd+bwd at every N; compute-only per rank: profiles/r05_rank_emulation.txt)")
    print("   N=1  1x1: 121.4 ms per iteration = %5.0f TFLOP/s (measured: the driver's command on one box)" % (F / 121.4 * 1e3))
    # N = 2, ulysses 2: per rank q 256 MiB, k / v 32 MiB each before the exchange; half of everything goes to the peer
    comp = 60.09
    ex = dict(fi=ms((128 + 32) * MiB), fo=ms(128 * MiB), bi=ms(128 * MiB), bo=ms((128 + 32) * MiB))
    for ng, c in ((1, 59.83), (2, comp)):
        tf, _ = pipeline(ex["fi"], c * 0.235, ex["fo"], ng)
        tb, _ = pipeline(ex["bi"], c * 0.765, ex["bo"], ng)
        tot = tf + tb
        print("   N=2  2x1, %d head group(s): exchanges %.2f + %.2f + %.2f + %.2f ms over ONE link, compute %.1f ms -> %.1f ms per iteration = %5.0f "
              "TFLOP/s on 2 GPUs, overlap %.2f" % (ng, ex["fi"], ex["fo"], ex["bi"], ex["bo"], c, tot, F / tot * 1e3,
                                                 1 - (tot - c) / sum(ex.values())))
    # ... with the self-chunk start (USP_SELF_CHUNK=1, built in round 5 for this grid: the first group's kernels start on the rank's
    # own rows, the first exchange of each pass runs beside them; compute-only cost of the split: none, profiles/r05_self_chunk.txt)
    tot = comp + ex["fo"] / 2 + ex["bo"] / 2
    print("   N=2  2x1, 2 head groups + self-chunk start: first-in exchanges hidden, last-out %.2f + %.2f ms exposed -> %.1f ms per iteration = %5.0f "
          "TFLOP/s on 2 GPUs, overlap %.2f" % (ex["fo"] / 2, ex["bo"] / 2, tot, F / tot * 1e3, 1 - (tot - comp) / sum(ex.values())))
    # N = 4, ring 4 zigzag: K / V of a peer 2 x 16 MiB (forward and again in the backward: three links in parallel); the travelling
    # fp32 dK + dV 64 MiB per hop, every hop but the last beside a step's kernels (5.6 ms), the last one rounded (32 MiB)
    comp, kv, hop = 29.54, ms(32 * MiB), ms(64 * MiB)
    step_b = comp * 0.765 / 4
    tot = comp + max(0.0, kv - comp * 0.235 / 4) + max(0.0, hop - step_b) * 3 + hop / 2
    comm = 2 * kv + 3 * hop + hop / 2
    print("   N=4  1x4 zigzag: K/V fetch %.2f ms x 2 (3 links in parallel), dK/dV hop %.2f ms against %.1f ms of kernels per step, last hop "
          "rounded %.2f ms exposed -> %.1f ms per iteration = %5.0f TFLOP/s on 4 GPUs, overlap %.2f"
          % (kv, hop, step_b, hop / 2, tot, F / tot * 1e3, 1 - (tot - comp) / comm))
    # N = 8: the section above with round 5's compute-only iteration (16.36 ms pipelined, 16.19 ms one exchange; round 2: 16.7)
    fwd, bwd = 16.19 * 0.235, 16.19 * 0.765
    e8 = dict(fi=ms(40 * MiB), fo=ms(32 * MiB), bi=ms(32 * MiB), bo=ms(40 * MiB))
    hop8, kv8 = ms(32 * MiB), ms(16 * MiB)
    for label, ng, eff, relay in (("one exchange (USP_SAFE_COMM=1)", 1, 1.0, 1.0), ("2 head groups pipelined (default)", 2, 16.19 / 16.36, 1.0),
                                  ("default + pair exchanges striped over 6 helpers (USP_EXCHANGE_RELAY=1)", 2, 16.19 / 16.36, 3 / 8)):
        e = {n: t * relay for n, t in e8.items()}
        tf, _ = pipeline(e["fi"], fwd, e["fo"], ng, eff)
        tb, _ = pipeline(e["bi"], bwd, e["bo"], ng, eff)
        last = hop8 / ng / 2
        tot = tf + tb + last
        comm = sum(e.values()) + 3 * hop8 + ng * last + 2 * kv8
        print("   N=8  2x4, %-74s %.1f ms per iteration = %5.0f TFLOP/s on 8 GPUs, overlap %.2f"
              % (label + ":", tot, F / tot * 1e3, 1 - (tot - (fwd + bwd) / eff) / comm))
        if ng == 2:
            # ... + the self-chunk start beside the ring (USP_SELF_CHUNK=1: step 0 of the first group's ring schedule starts on the
            # owned zigzag chunk): the first-in exchange of each pass runs beside 1/4 of that step's kernels on the slower rank
            # (forward: 0.25 x 0.496 ms causal launch of an 8-head group; backward: 0.25 x (0.849 + 0.605) ms -- rank trace,
            # profiles/r05_rank_emulation.txt); what the exchange takes beyond that stays exposed
            hid = min(e["fi"] / ng, 0.25 * 0.496) + min(e["bi"] / ng, 0.25 * (0.849 + 0.605))
            print("   N=8  2x4, %-74s %.1f ms per iteration = %5.0f TFLOP/s on 8 GPUs, overlap %.2f"
                  % ("  ... + self-chunk start beside the ring (USP_SELF_CHUNK=1):", tot - hid, F / (tot - hid) * 1e3,
                     1 - (tot - hid - (fwd + bwd) / eff) / comm))
    print("   (overlap >= 0.90 at N=8 needs < 0.46 ms exposed of 4.6 ms: the first-in / last-out exchange of each pass alone is 1.18 ms at 64 GB/s;"
          "\n    the self-chunk start hides the first-in part as far as a quarter of step 0 lasts; the last-out part needs row-chunked tails"
          "\n    -- DESIGN.md 5: not built)")


def round6(gbs_list=(48.0, 64.0, 96.0), comp=None):
    """Round 6: the 8-GPU grid's iteration replayed launch by launch on two resources (the compute stream, the Ulysses lane) for the
    schedules the library can run, at several assumed link rates -- the decision "tails on by default" must not hang on one rate.
    Per head group (8 query heads, 1 KV head) of a rank, kernel times from the one-GPU rank trace (profiles/r05_rank_emulation.txt,
    r06_rank_emulation.txt): a ring step of the forward = 2 c^2 score entries per head = `fstep` ms, of the backward `bstep` ms
    (dK/dV launch 0.58 of it, dQ launch 0.42); the forward's launches behind step 0 by ring rank r (grouped mesh fetch, one wave):
    r0: 3 c^2 + 3 c^2, r1: 2 + 2 + 2, r2: 4 + 1 + 1, r3: 6 (x c^2) -- the LAST one finalises the rows that travel.
    `comp` = the compute-only iteration (ms) of the schedule with and without the round-6 launches."""
    # compute-only iteration of the SLOWEST rank (ring rank 2) per schedule, one box, alternating (profiles/r06_rank_emulation.txt);
    # ring rank 0 on the same box: 14.30 / 14.60 / 14.79 / 14.64 / 14.92
    comp = comp or {"r5": 15.72, "sc": 15.98, "sc_t4": 16.15, "sc_t2": 16.08, "all_t4": 16.29}
    # N = 2 (ulysses 2, ring degree 1): two head groups; compute-only iteration of a rank by schedule, one box, alternating
    # (profiles/r06_rank_emulation_all_grids.txt): 60.79 (round 5) / 60.95 (+ self-chunk start) / 61.28 (+ tails in 4 pieces + dq first) /
    # 61.05 (2 pieces).  Per group: q|k|v in 80 MiB, out 64 MiB, dO in 64 MiB, dq 64 + dk|dv 16 MiB out, all over ONE link.
    print("\nround 6: the 2-GPU grid (ulysses 2, ring degree 1, same workload)")
    for gbs in gbs_list:
        ms = lambda nbytes: nbytes / (gbs * 1e9) * 1e3
        fi, fo, bi, bq, bkv = ms(80 * MiB), ms(64 * MiB), ms(64 * MiB), ms(64 * MiB), ms(16 * MiB)
        comm = 2 * (fi + fo + bi + bq + bkv)
        for name, c_iter, exposed in (("round-5 default", 60.79, fi + fo + bi + bq + bkv),
                                      ("+ self-chunk start (first-in exchanges hidden: the owned quarter lasts 1.8 / 5.8 ms)", 60.95, fo + bq + bkv),
                                      ("round-6 default: + tails in 4 pieces + dq first", 61.28, fo / 4 + bkv),
                                      ("... 2 pieces", 61.05, fo / 2 + bkv)):
            tot = c_iter + exposed
            print("  %3.0f GB/s: %-86s %.2f ms per iteration = %5.0f TFLOP/s on 2 GPUs, overlap %.2f"
                  % (gbs, name + ":", tot, 123.15 / tot * 1e3, 1 - exposed / comm))
    # N = 4 (ring 4 zigzag, no exchange): round 6 issues every backward step as dK/dV launch | hop | dQ launch, so the last hop
    # (rounded: 32 MiB) runs beside the last step's dQ launch (0.42 of a 5.6 ms step) instead of behind it
    print("\nround 6: the 4-GPU grid (ring 4 zigzag, same workload), compute-only 31.43 ms on the slowest ring rank (2), 28.4 on rank 0 -- the split costs nothing (profiles/r06_rank_emulation_all_grids.txt)")
    for gbs in gbs_list:
        ms = lambda nbytes: nbytes / (gbs * 1e9) * 1e3
        comp4, kv, hop = 31.43, ms(32 * MiB), ms(64 * MiB)
        step_b = comp4 * 0.765 / 4
        comm = 2 * kv + 3 * hop + hop / 2
        for name, last in (("round 5 (one call per step)", hop / 2), ("round 6 (steps split around their hop)", max(0.0, hop / 2 - 0.42 * step_b))):
            tot = comp4 + max(0.0, kv - comp4 * 0.235 / 4) + max(0.0, hop - step_b) * 3 + last
            print("  %3.0f GB/s: %-42s %.2f ms per iteration = %5.0f TFLOP/s on 4 GPUs, overlap %.2f" % (gbs, name + ":", tot, 123.15 / tot * 1e3, 1 - (tot - comp4) / comm))
    print("\nround 6: the 8-GPU grid (ulysses 2 x ring 4, B1 S65536 H32/Hkv4 fwd+bwd), slowest ring rank, by link rate")
    last_launch = {0: 1.5, 1: 1.0, 2: 0.5, 3: 3.0}            # final forward launch of a group, in ring steps
    for gbs in gbs_list:
        ms = lambda nbytes: nbytes / (gbs * 1e9) * 1e3
        fi, fo, bi, bq, bkv, hop = ms(20 * MiB), ms(16 * MiB), ms(16 * MiB), ms(16 * MiB), ms(4 * MiB), ms(8 * MiB)
        kvf = ms(8 * MiB)                                      # K/V of a peer, one head group (three links in parallel)
        comm = 2 * (fi + fo + bi + bq + bkv) + 2 * 3 * ms(16 * MiB) + 2 * hop + 4 * kvf     # every transfer once (fp32 hops: 16 MiB)
        rows = []
        for name, own_all, own_first, n_tail, dq_first, c_iter in (
                ("round-5 default (USP_SELF_CHUNK=0 USP_TAILS=0)", False, False, 0, False, comp["r5"]),
                ("+ self-chunk start of the first group (round 5 opt-in; USP_TAILS=0)", False, True, 0, False, comp["sc"]),
                ("round-6 default: self-chunk start + tails in 4 row pieces + dq first", False, True, 4, True, comp["sc_t4"]),
                ("... with 2 pieces (USP_TAILS=2)", False, True, 2, True, comp["sc_t2"]),
                ("... every group's own chunk first (USP_SELF_CHUNK=all), 4 pieces", True, True, 4, True, comp["all_t4"])):
            worst = 0.0
            for r in range(4):
                fstep, bstep = c_iter * 0.235 / 8, c_iter * 0.765 / 8           # per group and ring step
                own = 0.164                      # the owned chunk's launch + its K-split merge (profiles/r06_rank_launches.txt)
                # ---- forward
                land = [fi, 2 * fi]
                t = (2 * own if own_all else (own if own_first else 0.0))
                lane = 2 * fi
                ends = []
                for g in range(2):
                    started_own = own_all or (own_first and g == 0)
                    t = max(t, land[g]) + 4 * fstep - (own if started_own else 0.0)
                    ends.append(t)
                L = last_launch[r] * fstep
                if n_tail:
                    Lp = L / n_tail                                             # (the pieces' K split + merge launches are in c_iter)
                    start = ends[1] - L
                    lane = max(lane, ends[0]) + fo                              # group 0's output
                    tt = start
                    for j in range(n_tail):
                        tt += Lp
                        lane = max(lane, tt) + fo / n_tail
                    fwd_end = lane
                else:
                    lane = max(lane, ends[0]) + fo
                    fwd_end = max(lane, ends[1]) + fo
                # ---- backward
                land = [bi, 2 * bi]
                lane = 2 * bi
                t = 0.0
                own_b = 0.444                    # delta + dK/dV + dQ (+ reduces) on the owned rows
                t = (own_b if own_first else 0.0)
                t = max(t, land[0]) + 4 * bstep - (own_b if own_first else 0.0)
                lane = max(lane, t + hop) + bq + bkv                            # group 0: hop pending on the lane, one exchange
                t = max(t, land[1]) + 4 * bstep
                if dq_first:
                    dq_done = t - 0.58 * bstep                                  # the dQ launch of the last step ends here
                    lane_q = max(lane, dq_done) + bq
                    bwd_end = max(lane_q, t + hop) + bkv
                else:
                    bwd_end = max(lane, t + hop) + bq + bkv
                worst = max(worst, fwd_end + bwd_end)
            rows.append((name, worst, c_iter))
        print("  %3.0f GB/s per link: transfers %.2f ms per rank and iteration" % (gbs, comm))
        for name, tot, c_iter in rows:
            print("     %-82s %6.2f ms = %5.0f TFLOP/s on 8 GPUs, compute-only %.2f ms, overlap %.2f"
                  % (name + ":", tot, 123.15 / tot * 1e3, c_iter, 1 - (tot - c_iter) / comm))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--link-gbs", type=float, default=64.0, help="one xGMI link, one direction, GB/s")
    a = ap.parse_args()
    bw = a.link_gbs * 1e9
    ms = lambda nbytes: nbytes / bw * 1e3
    print(f"assumed link rate {a.link_gbs:.0f} GB/s per direction; compute-only times from profiles/r02_rank_emulation.txt\n")

    # configs[2]: 2 GPUs, ulysses 2, B1 S16384 H16 D128 bf16, forward.  Per rank: q,k,v local 3 x 32 MiB, half goes to the peer.
    t_in, t_out = ms(48 * MiB), ms(16 * MiB)
    print("configs[2]  2 GPUs  ulysses 2, forward          exchange in %.2f ms, out %.2f ms over ONE link" % (t_in, t_out))
    for ng, comp in ((1, 0.48 / 0.93), (2, 2 * 0.30), (4, 4 * 0.26)):      # kernel time of the group launches (kbench)
        tot, exp = pipeline(t_in, comp, t_out, ng)
        print("   %d head group(s): %.2f ms per iteration = %5.0f TFLOP/s on 2 GPUs (attention %.2f ms, exposed %.2f ms)"
              % (ng, tot, 2 * 0.2749 / tot * 1e3, comp, exp))
    # with the forward K split (usp_fwd_args.k_splits; kbench ksplit, profiles/r02_kbench_ksplit.log): a 4-head group
    # 0.257 ms at n = 2 (0.345 unsplit on that box), a 2-head group 0.147 ms at n = 4 (0.247 unsplit)
    for ng, comp in ((2, 2 * 0.257), (4, 4 * 0.147)):
        tot, exp = pipeline(t_in, comp, t_out, ng)
        print("   %d head group(s), K split: %.2f ms per iteration = %5.0f TFLOP/s on 2 GPUs (attention %.2f ms, exposed %.2f ms)"
              % (ng, tot, 2 * 0.2749 / tot * 1e3, comp, exp))

    # configs[3]: 4 GPUs, ring 4 zigzag, B1 S32768 H16, forward.  K/V 2 x 32 MiB per peer, each peer over its own link.
    # kernel times of a rank (rocprofv3, profiles/r02_rank_emulation.txt): step 0 (causal) 0.294 ms, a half step 0.143 ms
    step0, half = 0.294, 0.143
    whole = ms(64 * MiB)                                                    # K and V of one peer over one link
    print("\nconfigs[3]  4 GPUs  ring 4 zigzag, forward      K/V of a peer: %.2f ms per link" % whole)
    t = step0
    for s in (1, 2, 3):                                                     # hop s has crossed s links one after the other
        t = max(t, s * whole) + 2 * half
    print("   hop-by-hop relay (reference order):  %.2f ms per iteration" % t)
    print("   one-wave mesh fetch:                 %.2f ms (steps 1-3 wait for the whole fetch)" % (max(step0, whole) + 6 * half))
    def waves(W, order, grouped=False):
        """Slowest rank's iteration with 2 W waves (W row ranges per K/V half; wave w lands at (w + 1) / (2 W) of the
        transfer).  order "wave": every launch that needs only the landed waves first (what the package issues);
        "step": a step's launches together (round 2's first form: rank 0 waits for the LAST wave at its first step).
        grouped (batch 1): the source ranks of a wave that share a query range are ONE launch, one merge epilogue."""
        # every q row x piece, or q[c:] x piece; each launch pays its own merge epilogue (read + write of the fp32
        # running output of its q rows), which the measured (half) step time holds once.  Measured on MI355X
        # (`kbench pieces`, profiles/r02_kbench_pieces*.log): +16.4 us per extra q[c:] launch, +39-41 us per extra
        # all-rows launch at this shape
        m_all, m_back = 0.040, 0.0165
        worst = 0.0
        for r in range(4):
            land = [(w + 1) * whole / (2 * W) for w in range(2 * W)]
            reads = lambda w, s: w < W or s > r
            if grouped:      # (wave, [steps]) launches: steps 1..r with every q row (front waves), steps r+1..3 with q[c:]
                seq = []
                for w in range(2 * W):
                    if w < W and r >= 1:
                        seq.append((w, list(range(1, r + 1)), True))
                    if r < 3:
                        seq.append((w, list(range(r + 1, 4)), False))
            elif order == "wave":
                seq = [(w, [s], s <= r) for w in range(2 * W) for s in (1, 2, 3) if reads(w, s)]
            else:
                seq = [(w, [s], s <= r) for s in (1, 2, 3) for w in range(2 * W) if reads(w, s)]
            t = step0
            for w, steps, all_rows in seq:
                base, merge = (2 * half, m_all) if all_rows else (half, m_back)
                t = max(t, land[w]) + len(steps) * (base - merge) / W + merge
            worst = max(worst, t)
        return worst
    print("   two-wave fetch, launches step by step: %.2f ms (slowest rank)" % waves(1, "step"))
    print("   two-wave fetch, launches wave by wave: %.2f ms (front halves land after %.2f ms)" % (waves(1, "wave"), whole / 2))
    print("   four waves (2 row ranges per half):    %.2f ms   [eight waves: %.2f ms]" % (waves(2, "wave"), waves(4, "wave")))
    worst = waves(2, "wave", True)
    print("   four waves, one launch per query range: %.2f ms   [default at this size; two waves %.2f, eight waves %.2f ms]"
          % (worst, waves(1, "wave", True), waves(4, "wave", True)))
    print("   -> %.0f TFLOP/s on 4 GPUs at the default, %.0f with the relay" % (4 * 1.0995 / worst * 1e3, 4 * 1.0995 / (3 * whole + 2 * half) * 1e3))

    # configs[4]: 8 GPUs, ulysses 2 x ring 4, B1 S65536 H32/4, forward + backward.
    fwd, bwd = 16.7 * 0.29, 16.7 * 0.71
    ex = dict(fi=ms(40 * MiB), fo=ms(32 * MiB), bi=ms(32 * MiB), bo=ms(40 * MiB))
    hop = ms(32 * MiB)                                                      # fp32 dK + dV of both KV heads, one hop
    kv = ms(16 * MiB)                                                       # K/V of one peer, both KV heads (mesh fetch: 3 links in parallel)
    print("\nconfigs[4]  8 GPUs  ulysses 2 x ring 4, fwd+bwd  exchanges %.2f + %.2f + %.2f + %.2f ms, dK/dV hop %.2f ms, K/V fetch %.2f ms x 2"
          % (ex["fi"], ex["fo"], ex["bi"], ex["bo"], hop, kv))

    def c5(ng, eff_f, eff_b, defer_tail, hop16, relay=1.0):
        """One iteration of a rank.  Exposed: the first input and the last output exchange of each pass (pipeline()), and
        the last dK/dV hop of a head group -- round 2: of EVERY group (the compute stream waited for it before the next
        group's kernels); round 3 (`defer_tail`): only of the last group (the hop is waited for on the exchange lane),
        at half the bytes when it travels rounded (`hop16`).  t_comm: every transfer once."""
        e = {n: t * relay for n, t in ex.items()}     # relay: the pair exchange striped over k helpers takes 3 / (k + 2)
        tf, _ = pipeline(e["fi"], fwd, e["fo"], ng, eff_f)
        tb, _ = pipeline(e["bi"], bwd, e["bo"], ng, eff_b)
        last = (hop / ng) * (0.5 if hop16 else 1.0)
        tot = tf + tb + (1 if defer_tail else ng) * last
        comm = sum(e.values()) + 3 * hop + ng * last + 2 * kv
        return tot, 1 - (tot - (fwd / eff_f + bwd / eff_b)) / comm
    for label, args in (("round 2: 1 head group (USP_PIPELINE_ULYSSES=0 / USP_SAFE_COMM=1)", (1, 1.0, 1.0, False, False)),
                        ("round 2: 2 head groups pipelined", (2, 0.95, 0.99, False, False)),
                        ("round 3: 2 groups, last hop pending on the lane + rounded (default)", (2, 0.97, 0.995, True, True)),
                        ("round 3: 1 head group, last hop rounded (USP_SAFE_COMM=1)", (1, 1.0, 1.0, True, True)),
                        ("round 3: default + pair exchanges striped over 6 helpers (USP_EXCHANGE_RELAY=1)", (2, 0.97, 0.995, True, True, 3 / 8))):
        tot, ov = c5(*args)
        print("   %-72s %.1f ms per iteration = %5.0f TFLOP/s on 8 GPUs; overlap = 1 - (t - t_compute)/t_comm = %.2f"
              % (label + ":", tot, 8 * 15.39 / tot * 1e3, ov))
    # What 0.90 would take (costed, not built -- DESIGN.md 5): what stays exposed with two groups is the first input and
    # the last output exchange of each pass (0.33 + 0.26 + 0.26 + 0.33 ms at 64 GB/s) and the last hop (0.13).
    e = dict(F1=ex["fi"] / 2, F2=ex["fo"] / 2, B1=ex["bi"] / 2, B2=ex["bo"] / 2, H=hop / 4)
    print("   exposed with the round-3 default: first-in fwd %.2f, last-out fwd %.2f, first-in bwd %.2f, last-out bwd %.2f, last hop %.2f = %.2f ms"
          % (e["F1"], e["F2"], e["B1"], e["B2"], e["H"], sum(e.values())))
    cut = dict(F1=0.10, F2=e["F2"] / 2, B1=0.0, B2=0.20, H=e["H"])   # self-chunk start, row-chunked tails: see DESIGN.md 5
    comm = sum(ex.values()) + 3 * hop + 2 * hop / 4 + 2 * kv
    print("   with self-chunk starts + row-chunked tails (3-4 launches instead of 1 at four places, ~0.2 ms of kernel time): %.2f ms exposed -> overlap %.2f"
          % (sum(cut.values()), 1 - sum(cut.values()) / comm))

    round5(ms)
    round6()

    # Ring backward: the travelling dK/dV (relay, the reference's order) against USP_DKDV_RETURN=direct (every block
    # straight to its owner over its own link, front-half blocks at half size).  Per ring rank; t_c = kernels of one step.
    print("\nring backward, dK/dV transport (ring 4):")
    for name, t_c, hop_mib in (("configs[4] rank, one of two head groups (GQA: 1 KV head)", 16.7 * 0.71 / 2 / 4, 16),
                               ("an MHA ring: configs[3]'s shape trained (16 KV heads, 8192 tokens per rank)", 2.5 * 2 * half, 128)):
        hop = ms(hop_mib * MiB)
        relay = t_c + sum(max(t_c, hop) for _ in range(3)) + hop          # hop s needs hop s-1 AND the kernels of step s
        t, landed = t_c, 0.0
        for s in (1, 2, 3):                                               # worst rank (0): every arriving block is whole
            t += t_c
            landed = max(landed, t + hop)                                 # own link per step: transfers overlap each other
        print("   %s:\n      kernels %.2f ms per step, fp32 dK+dV %d MiB = %.2f ms per link -> relay %.1f ms, direct %.1f ms per backward"
              % (name, t_c, hop_mib, hop, relay, landed))


if __name__ == "__main__":
    main()
