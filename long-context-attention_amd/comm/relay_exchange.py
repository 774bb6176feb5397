"""The Ulysses exchange of a PAIR of ranks striped over the idle links of the xGMI mesh (round 3; off by default).

At ulysses degree 2 the head all-to-all is one transfer to one peer: it rides ONE of the seven point-to-point links of a
GPU while the other six idle -- and it is exposed exactly when they idle, at the start and the end of a pass (first
input / last output exchange of the head-group pipeline).  Here the peer's chunk is cut along its rows
into one DIRECT stripe and one stripe per HELPER (every other rank of the sequence-parallel block); two grouped send/recv
calls on the world group move it:
    phase 1: the direct stripe goes to the peer, stripe j to helper j (and this rank receives, as a helper, one stripe from
             every rank outside its own pair);
    phase 2: every helper forwards the stripe it holds to the peer of the rank it came from.
The direct stripe is twice a helper stripe, so both phases take about one helper-stripe time on every link: with k helpers
the exchange takes 3 / (k + 2) of the direct transfer (k = 6 on an 8-GPU node: 0.375), at the price of the relayed bytes
crossing two links.  Every rank posts the same two grouped calls in the same order, whatever its pair.

Enabled by USP_EXCHANGE_RELAY=1 (or hybrid.async_attn_layer._COMM_OVERRIDE["relay"]), for ulysses degree 2 on the grid
`set_seq_parallel_pg` built, with at least two helpers.  bench.py --gpus 8 tries it as a third, deadline-guarded mode.
Results are bit-identical to `all_to_all_single` (the same bytes land in the same places)."""
import os

import torch
import torch.distributed as dist

_OVERRIDE = {}            # {"relay": bool} -- bench.py / tests, instead of the environment
GRID = None               # (ulysses degree, ring degree, world size, use_ulysses_low), recorded by set_seq_parallel_pg


def relay_enabled() -> bool:
    if "relay" in _OVERRIDE:
        return bool(_OVERRIDE["relay"])
    return os.environ.get("USP_EXCHANGE_RELAY", "0") == "1"


def pair_and_helpers(rank: int, grid=None):
    """(peer of `rank`, helpers = the other ranks of its sequence-parallel block, ascending) on the recorded grid, or None
    when the relay does not apply (ulysses degree != 2, fewer than two helpers, no grid)."""
    grid = GRID if grid is None else grid
    if grid is None:
        return None
    ud, rd, _ws, low = grid
    sp = ud * rd
    if ud != 2 or sp < 4:
        return None
    base = (rank // sp) * sp
    local = rank - base
    peer = base + ((local ^ 1) if low else (local + rd) % sp)      # globals.py: contiguous pairs, or stride rd
    return peer, [w for w in range(base, base + sp) if w not in (rank, peer)]


def stripe_rows(rows: int, k: int):
    """(direct rows, rows per helper): the direct stripe is twice a helper's (it needs no second phase)."""
    r = rows // (k + 2)
    return rows - k * r, r


def _offsets(sender: int, peer: int, helpers, d: int, r: int):
    """Row offset of the stripe `sender` assigns to each destination: destinations in ascending rank order."""
    off, pos = {}, 0
    for w in sorted(helpers + [peer]):
        off[w] = pos
        pos += d if w == peer else r
    return off


def applicable(send: torch.Tensor, group) -> bool:
    if not relay_enabled() or GRID is None or send.shape[0] != 2 or dist.get_world_size(group) != 2:
        return False
    plan = pair_and_helpers(dist.get_rank())
    return plan is not None and stripe_rows(send.shape[1], len(plan[1]))[1] > 0


_AGREED = set()          # (shape, dtype) signatures every rank of THIS sequence-parallel block has confirmed


def forget_agreements():
    """set_seq_parallel_pg: a re-initialised grid starts without confirmed signatures (another block, other peers)."""
    _AGREED.clear()


def _agree_once(send: torch.Tensor):
    """First use of a buffer signature: every rank of the SEQUENCE-PARALLEL BLOCK (the ranks the two grouped phases below
    couple: base .. base + ud * rd -- not the world: data-parallel replicas may legitimately exchange other shapes, or reach
    a new shape at another step) tells every other one its signature, point to point, in ONE grouped send/recv, and raises
    when they differ (the phases would otherwise pair up buffers of different sizes).  A rank of the block that took the
    direct path instead leaves the others waiting here -- as it would in phase 1 -- which is why USP_EXCHANGE_RELAY must be
    set for the whole block or not at all."""
    sig = (tuple(send.shape), str(send.dtype))
    if sig in _AGREED:
        return
    h = 0
    for x in (*send.shape, send.element_size()):
        h = (h * 1000003 + int(x)) % (1 << 40)
    me = dist.get_rank()
    peer, helpers = pair_and_helpers(me)
    block = sorted(helpers + [peer])
    mine = torch.tensor([float(h)], dtype=torch.float64, device=send.device)
    theirs = torch.zeros((len(block), 1), dtype=torch.float64, device=send.device)
    ops = []
    for j, w in enumerate(block):
        ops += [dist.P2POp(dist.isend, mine, w), dist.P2POp(dist.irecv, theirs[j], w)]
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    if not bool((theirs == float(h)).all().item()):
        raise RuntimeError(f"relayed pair exchange: the ranks of this sequence-parallel block disagree on the exchanged buffer "
                           f"(here {sig}); USP_EXCHANGE_RELAY needs every rank of the block in the same exchange with the same shape")
    _AGREED.add(sig)


def exchange_relayed(send: torch.Tensor, group) -> torch.Tensor:
    """all_to_all_single(send) for a pair: send (2, rows, ...) contiguous, chunk p goes to pair rank p."""
    _agree_once(send)
    me, me_g = dist.get_rank(), dist.get_rank(group)
    peer, helpers = pair_and_helpers(me)
    assert dist.get_global_rank(group, 1 - me_g) == peer, "the recorded grid does not match the ulysses group"
    recv = torch.empty_like(send)
    recv[me_g].copy_(send[me_g])                                  # the self chunk (all_to_all_single copies it too)
    rows = send.shape[1]
    d, r = stripe_rows(rows, len(helpers))
    out_c, in_c = send[1 - me_g].reshape(rows, -1), recv[1 - me_g].reshape(rows, -1)
    mine, theirs = _offsets(me, peer, helpers, d, r), _offsets(peer, me, helpers, d, r)
    stage = torch.empty((len(helpers), r, out_c.shape[1]), dtype=send.dtype, device=send.device)
    # phase 1: my stripes out; the peer's direct stripe in (straight into its place); one stripe of every other pair's
    # traffic in (I am their helper): helper x's own chunk assigns me the rows at ITS offset for destination `me`
    ops = [dist.P2POp(dist.isend, out_c[mine[peer]:mine[peer] + d], peer),
           dist.P2POp(dist.irecv, in_c[theirs[me]:theirs[me] + d], peer)]
    for j, x in enumerate(helpers):
        ops += [dist.P2POp(dist.isend, out_c[mine[x]:mine[x] + r], x), dist.P2POp(dist.irecv, stage[j], x)]
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    # phase 2: forward what I hold to the peer of its sender; the stripes of MY peer's chunk arrive from the helpers
    ops = []
    for j, x in enumerate(helpers):
        ops += [dist.P2POp(dist.isend, stage[j], pair_and_helpers(x)[0]),
                dist.P2POp(dist.irecv, in_c[theirs[x]:theirs[x] + r], x)]
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    return recv
