"""N > 1 on ONE GPU: several processes share cuda:0 and talk over gloo (RCCL refuses two ranks on one
device), running the package's real path -- HIP kernels, KVRelay side stream + events, all-to-all
pack/unpack kernels, autograd -- against the reference's golden runs.  What this adds over
tests/test_dist_cpu.py (CPU tensors, oracle block kernel) is the device-side ordering between the compute
stream, the relay stream and the transport; what it cannot cover is RCCL itself.

gloo's point-to-point path is NOT stream-ordered for device tensors: its threads read / write the device
buffer directly from the CPU as soon as the call is posted, where RCCL enqueues the transfer behind the
work already on the stream.  `_order_p2p_like_rccl` therefore makes every RingComm.commit wait for the
device first (a send posted right after the kernel that produces its payload would otherwise race with
that kernel -- observed on the travelling dK/dV accumulators at world size 8).  With that, the ordering
these tests check is the package's own: the relay stream vs the compute stream, buffer reuse across
iterations, and the all-to-all pack / unpack kernels between processes."""
import os
import numpy as np
import pytest
import torch

from dist_util import run_distributed
from golden_util import Golden, TOL, VarlenGolden, assert_close, golden_files, varlen_golden_files

pytestmark = pytest.mark.gpu


def _order_p2p_like_rccl():
    import yunchang_amd.ring.utils as U
    if getattr(U.RingComm, "_gloo_ordered", False):
        return
    orig = U.RingComm.commit

    def commit(self):
        torch.cuda.synchronize()
        return orig(self)

    U.RingComm.commit = commit
    U.RingComm._gloo_ordered = True


def _usp_gpu_worker(rank, ws, path, pipelined=False):
    import yunchang_amd as Y
    _order_p2p_like_rccl()
    if pipelined:        # head groups pipelined on the "ulysses" side stream, ALSO beside a ring (USP_PIPELINE_ULYSSES=1;
        import os        # the default at ring degree 1); the fixtures are tiny, so let the groups form anyway
        import yunchang_amd.hybrid.async_attn_layer as AL
        assert "USP_PACK_QKV" not in os.environ and "USP_SAFE_COMM" not in os.environ
        os.environ["USP_PIPELINE_ULYSSES"] = "1"
        AL._FILL_ITEMS = 1
    from yunchang_amd.kernels import get_block_backend
    assert get_block_backend().name == "hip"
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    g = Golden(path)
    dtype = getattr(torch, g.dtype)
    Y.set_seq_parallel_pg(g.ud, g.rd, rank, ws)
    ext = Y.EXTRACT_FUNC_DICT[g.impl]
    glob = [torch.from_numpy(t).to(dtype) for t in (g.q, g.k, g.v, g.dout)]
    lq, lk, lv, ldo = (ext(t, rank, world_size=ws, rd=g.rd, ud=g.ud).detach().clone().to(dev) for t in glob)
    kw = dict(dropout_p=0, causal=g.causal, window_size=(-1, -1), softcap=0.0, alibi_slopes=None,
              deterministic=False, return_attn_probs=True)
    qkv = None
    if g.layer == "ulysses":
        attn = Y.UlyssesAttention(Y.PROCESS_GROUP.ULYSSES_PG, attn_type=Y.AttnType.HIP)
    elif g.layer == "hybrid":
        attn = Y.LongContextAttention(ring_impl_type=g.impl, attn_type=Y.AttnType.HIP)
        assert (attn._packed_exchange(lq, lk) is not None) == (g.ud > 1)      # one packed q|k|v exchange
    else:                # LongContextAttentionQKVPacked + SeqAllToAll5D
        attn = Y.LongContextAttentionQKVPacked(ring_impl_type=g.impl, attn_type=Y.AttnType.HIP)
        qkv = torch.stack([lq, lk, lv], dim=2).requires_grad_(True)
    res = {}
    for it in range(2):                      # twice: buffers / streams must be reusable
        if qkv is not None:
            qkv.grad = None
            out = attn(qkv, **kw)
        else:
            for t in (lq, lk, lv):
                t.requires_grad_(True)
                t.grad = None
            out = attn(lq, lk, lv, **kw)
        out.backward(ldo)
        torch.cuda.synchronize()
        grads = (qkv.grad[:, :, 0], qkv.grad[:, :, 1], qkv.grad[:, :, 2]) if qkv is not None else \
                (lq.grad, lk.grad, lv.grad)
        res = dict(out=out.detach().float().cpu().numpy(), dq=grads[0].float().cpu().numpy(),
                   dk=grads[1].float().cpu().numpy(), dv=grads[2].float().cpu().numpy())
    return res


def _batch2_worker(rank, ws, ud, rd, impl, Hq, Hkv, D, self_chunk=False, tails=None):
    """No fixture has batch > 1 on a ulysses x ring grid: exact attention (fp64 oracle) on the unsharded
    tensors is the reference here.  Batch 2 is what gives the seq-major views the exchange hands the ring a
    real batch stride (the P > 1 gradient cast used to reject them)."""
    import yunchang_amd as Y
    import yunchang_amd.hybrid.async_attn_layer as AL
    from oracle import usp_oracle as O
    _order_p2p_like_rccl()
    AL._FILL_ITEMS = 1
    calls = []
    AL._COMM_OVERRIDE["self_chunk"] = "1" if self_chunk else "0"       # (the default since round 6: pinned either way)
    pieces = []
    if tails:            # row-chunked tails with an explicit piece count (the override skips the fill cap of tails_mode)
        AL._COMM_OVERRIDE["tails"] = str(tails)
        import yunchang_amd.comm.all_to_all as A_
        real_pack = A_.pack_seq_rows
        A_.pack_seq_rows = lambda *a: (pieces.append(a[2:]), real_pack(*a))[1]
    if self_chunk and rd == 1:
        real_f, real_b = AL._split_first_forward, AL._split_first_backward
        AL._split_first_forward = lambda *a: (calls.append("f"), real_f(*a))[1]
        AL._split_first_backward = lambda *a: (calls.append("b"), real_b(*a))[1]
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    Y.set_seq_parallel_pg(ud, rd, rank, ws)
    torch.manual_seed(0)
    rows = 64 if not (self_chunk or tails) else (333 * 2 if not tails else (99 * 2 if ws <= 4 else 65 * 2))
    B, S = 2, rows * ws                  # (with tails: smaller, the fp64 truth costs S^2 per process and the GPU suite has a time limit)
    q, k, v, do = (torch.randn(B, S, h, D).to(torch.bfloat16) for h in (Hq, Hkv, Hkv, Hq))
    ext = Y.EXTRACT_FUNC_DICT[impl]
    qn, kn, vn, don = (t.float().numpy().astype(np.float64) for t in (q, k, v, do))
    ro, rl = O.attention_ref(qn, kn, vn, causal=True)
    truth = [ext(torch.from_numpy(np.ascontiguousarray(t)), rank, world_size=ws, rd=rd, ud=ud).float().numpy()
             for t in (ro,) + tuple(O.block_bwd(don, qn, kn, vn, ro, rl, None, True))]
    lq, lk, lv, ldo = (ext(t, rank, world_size=ws, rd=rd, ud=ud).detach().clone().to(dev) for t in (q, k, v, do))
    for t in (lq, lk, lv):
        t.requires_grad_(True)
    out = Y.LongContextAttention(ring_impl_type=impl, attn_type=Y.AttnType.HIP)(lq, lk, lv, causal=True)
    out.backward(ldo)
    torch.cuda.synchronize()
    got = [t.detach().float().cpu().numpy() for t in (out, lq.grad, lk.grad, lv.grad)]
    if rd == 1:
        assert calls == (["f", "b"] if self_chunk else []), calls      # the split path ran (forward and backward), or did not
    if tails:
        assert len(pieces) == tails, pieces                            # the last group's output left in `tails` row pieces
    return got, truth


def _probe_worker(rank, ws):
    import torch.distributed as dist
    t = torch.full((4,), float(rank), device="cuda:0")
    if rank == 0:
        dist.send(t, 1)
    else:
        dist.recv(t, 0)
    torch.cuda.synchronize()
    return float(t[0])


def _gloo_moves_device_tensors():
    """gloo in this build must be able to send/recv device tensors; otherwise these tests cannot run."""
    try:
        return run_distributed(_probe_worker, 2) == [0.0, 0.0]
    except AssertionError:
        return False


@pytest.fixture(scope="module")
def gloo_cuda():
    if not _gloo_moves_device_tensors():
        pytest.skip("gloo cannot move device tensors in this build")


DENSE = [f for f in golden_files() if "_w1" not in f]


@pytest.mark.parametrize("path", DENSE, ids=lambda p: p.split("/")[-1][:-4])
def test_usp_multiprocess_one_gpu(gloo_cuda, path):
    g = Golden(path)
    res = run_distributed(_usp_gpu_worker, g.ws, path)
    for r in range(g.ws):
        assert_close(res[r]["out"], g.out[r], *TOL[g.dtype]["out"], f"{g.name} out rank {r}")
        for key in ("dq", "dk", "dv"):
            assert_close(res[r][key], getattr(g, key)[r], *TOL[g.dtype]["grad"], f"{g.name} {key} rank {r}")


PIPE = [f for f in DENSE if "c3_w2_u2r1" in f or "c5_w8_u2r4_gqa_bf16" in f or "n_w4_u2r2_strip" in f]


@pytest.mark.parametrize("path", PIPE, ids=lambda p: p.split("/")[-1][:-4])
def test_long_context_attention_with_pipelined_exchange(gloo_cuda, path):
    """LongContextAttention with the packed q|k|v exchange pipelined over head groups on the "ulysses" side stream,
    beside the ring relay on the "ring" side stream, kernels launched interleavable (USP_PIPELINE_ULYSSES=1; the
    default at ring degree 1), against the reference goldens (incl. ulysses 2 x ring 4, GQA)."""
    g = Golden(path)
    res = run_distributed(_usp_gpu_worker, g.ws, path, True)
    for r in range(g.ws):
        assert_close(res[r]["out"], g.out[r], *TOL[g.dtype]["out"], f"{g.name} out rank {r}")
        for key in ("dq", "dk", "dv"):
            assert_close(res[r][key], getattr(g, key)[r], *TOL[g.dtype]["grad"], f"{g.name} {key} rank {r}")


@pytest.mark.parametrize("ws,ud,rd,impl,Hq,Hkv,D", [(4, 2, 2, "zigzag", 8, 4, 128),
                                                    (4, 2, 2, "strip", 8, 2, 128), (4, 1, 4, "zigzag", 4, 2, 128),
                                                    (8, 4, 2, "zigzag", 8, 4, 128)] +      # ulysses 4: P = 4 exchange kernels
                         ([(4, 2, 2, "basic", 4, 4, 64)] if os.environ.get("USP_GPU_ALL_FIXTURES") == "1" else []))
def test_batch2_on_a_ulysses_x_ring_grid(gloo_cuda, ws, ud, rd, impl, Hq, Hkv, D):
    for got, truth in run_distributed(_batch2_worker, ws, ud, rd, impl, Hq, Hkv, D):
        for a, t, key in zip(got, truth, ("out", "dq", "dk", "dv")):
            assert_close(a, t, *TOL["bfloat16"]["out" if key == "out" else "grad"], f"B=2 {impl} {key}")


@pytest.mark.parametrize("impl,Hq,Hkv,D", [("zigzag", 4, 4, 128)] +       # (ring degree 1: "basic" takes the same split in the layer)
                         ([("basic", 8, 2, 128), ("basic", 4, 4, 64)] if os.environ.get("USP_GPU_ALL_FIXTURES") == "1" else []))
def test_self_chunk_start_two_processes_one_gpu(gloo_cuda, impl, Hq, Hkv, D):
    """USP_SELF_CHUNK=1 on the 2-GPU grid with the HIP kernels (two processes sharing the GPU): the first head group's block as
    two / three launches on views of the send and receive buffers (odd row counts: 666 rows per rank, the split at 666; merge-in
    with a partial final range on rank 1; fp32 dK / dV accumulated across the launches; GQA workspace path), batch 2, against
    exact attention and its gradients."""
    for got, truth in run_distributed(_batch2_worker, 2, 2, 1, impl, Hq, Hkv, D, True):
        for a, t, key in zip(got, truth, ("out", "dq", "dk", "dv")):
            assert_close(a, t, *TOL["bfloat16"]["out" if key == "out" else "grad"], f"self-chunk {impl} {key}")


@pytest.mark.parametrize("ws,rd,impl,Hq,Hkv,self_chunk,tails", [(4, 2, "zigzag", 8, 4, True, 3),      # beside a ring: K-cut pieces of the final launch
                                                                (8, 4, "zigzag", 8, 2, True, 2),      # the 8-GPU grid's shape of schedule, ONE head group
                                                                (2, 1, "basic", 8, 4, False, 3)])     # ring degree 1: the block in the layer
def test_row_chunked_tails_on_the_gpu(gloo_cuda, ws, rd, impl, Hq, Hkv, self_chunk, tails):
    """The round-6 tails with the HIP kernels on ragged sizes (198 rows per rank -- 130 on the 8-rank grid --: pieces of 66 / 33 / 65 rows, K cuts of the pieces
    beside a ring, merge-in with partial final ranges, dq rounded in the last step's dQ epilogue), batch 2, processes sharing
    the GPU, against exact attention and its gradients.  (The full-size RCCL virtual grids run the even default shapes.)"""
    for got, truth in run_distributed(_batch2_worker, ws, 2, rd, impl, Hq, Hkv, 128, self_chunk, tails):
        for a, t, key in zip(got, truth, ("out", "dq", "dk", "dv")):
            assert_close(a, t, *TOL["bfloat16"]["out" if key == "out" else "grad"], f"tails {impl} {key}")


def _varlen_gpu_worker(rank, ws, path):
    import torch.distributed as dist
    import yunchang_amd as Y
    _order_p2p_like_rccl()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    g = VarlenGolden(path)
    dtype = getattr(torch, g.dtype)
    layout = "zigzag" if g.impl == "zigzag" else "basic"
    lq, lk, lv, ldo = (Y.extract_local_varlen(torch.from_numpy(t).to(dtype), g.cu, rank, ws, layout).to(dev)
                       for t in (g.q, g.k, g.v, g.dout))
    for t in (lq, lk, lv):
        t.requires_grad_(True)
    cu_local = torch.tensor(g.cu_local, dtype=torch.int32, device=dev)
    fn = Y.zigzag_ring_flash_attn_varlen_func if g.impl == "zigzag" else Y.ring_flash_attn_varlen_func
    out, lse, _ = fn(lq, lk, lv, cu_local, g.max_local, causal=True, return_attn_probs=True,
                     group=dist.group.WORLD)
    out.backward(ldo)
    torch.cuda.synchronize()
    return dict(out=out.detach().float().cpu().numpy(), lse=Y.flatten_lse(lse.detach(), cu_local).cpu().numpy(),
                dq=lq.grad.float().cpu().numpy(), dk=lk.grad.float().cpu().numpy(),
                dv=lv.grad.float().cpu().numpy())


@pytest.mark.parametrize("path", varlen_golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_varlen_ring_multiprocess_one_gpu(gloo_cuda, path):
    g = VarlenGolden(path)
    res = run_distributed(_varlen_gpu_worker, g.ws, path)
    for r in range(g.ws):
        assert_close(res[r]["out"], g.out[r], *TOL[g.dtype]["out"], f"{g.name} out rank {r}")
        assert_close(res[r]["lse"], g.lse[r], *TOL[g.dtype]["out"], f"{g.name} lse rank {r}")
        for key in ("dq", "dk", "dv"):
            assert_close(res[r][key], getattr(g, key)[r], *TOL[g.dtype]["grad"], f"{g.name} {key} rank {r}")


RING_BWD = [f for f in DENSE if Golden(f).rd > 1 and Golden(f).bwd]
# (an opt-in transport, bit-identical to the relay on EVERY ring fixture on gloo -- tests/test_dist_cpu.py -- and on the GPU in rounds
# 3-5; the driver's GPU suite has a time limit, so by default two fixtures run here: the 8-GPU grid and a stripe ring.  USP_GPU_ALL_FIXTURES=1: all of them.)
if os.environ.get("USP_GPU_ALL_FIXTURES") != "1":
    RING_BWD = [f for f in RING_BWD if any(t in f for t in ("c5_w8_u2r4_gqa_bf16", "n_w4_u2r2_strip"))]


@pytest.mark.parametrize("path", RING_BWD, ids=lambda p: p.split("/")[-1][:-4])
def test_direct_dkdv_return_on_the_gpu(gloo_cuda, path, monkeypatch):
    """USP_DKDV_RETURN=direct through the HIP kernels and the real streams: the owner adds the arriving blocks in
    the relay's order, so the gradients must be bit-identical to the relay's (and match the reference's run)."""
    g = Golden(path)
    relay = run_distributed(_usp_gpu_worker, g.ws, path)
    monkeypatch.setenv("USP_DKDV_RETURN", "direct")
    direct = run_distributed(_usp_gpu_worker, g.ws, path)
    for r in range(g.ws):
        for key in ("dq", "dk", "dv"):
            assert np.array_equal(direct[r][key], relay[r][key]), f"{g.name} {key} rank {r}"
            assert_close(direct[r][key], getattr(g, key)[r], *TOL[g.dtype]["grad"], f"{g.name} {key} rank {r}")
