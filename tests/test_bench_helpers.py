"""CPU tests of bench.py's helper logic that the single-GPU runs never reach (N > 1)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from dist_util import run_distributed
from oracle import usp_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_workloads_are_the_metrics_configuration():
    """Every GPU count runs the configuration BASELINE.json's metric is quoted on (configs[4]'s global tensors, fwd+bwd) on the
    grid BASELINE names for that count; the per-count configs of rounds 1-4 stay selectable (USP_BENCH_WORKLOAD=configs)."""
    b = _bench()
    w = b.WORKLOADS
    for n, (ud, rd) in {1: (1, 1), 2: (2, 1), 4: (1, 4), 8: (2, 4)}.items():
        c = w[n]
        assert (c["B"], c["S"], c["Hq"], c["Hkv"], c["D"], c["bwd"]) == (1, 65536, 32, 4, 128, True), c
        assert (c["ud"], c["rd"]) == (ud, rd) and c["ud"] * c["rd"] == n
        assert c["Hq"] % ud == 0 and c["Hkv"] % ud == 0 and c["S"] % (2 * rd * ud) == 0
    assert w[4]["impl"] == w[8]["impl"] == "zigzag"
    k = b.CONFIG_WORKLOADS
    assert (k[1]["B"], k[1]["S"], k[1]["Hq"], k[1]["D"], k[1]["ud"], k[1]["rd"]) == (2, 8192, 16, 128, 1, 1) and b.C2 is k[1]
    assert (k[2]["S"], k[2]["ud"], k[2]["rd"]) == (16384, 2, 1)
    assert (k[4]["S"], k[4]["ud"], k[4]["rd"], k[4]["impl"]) == (32768, 1, 4, "zigzag")
    assert (k[8]["S"], k[8]["Hq"], k[8]["Hkv"], k[8]["ud"], k[8]["rd"], k[8]["bwd"]) == (65536, 32, 4, 2, 4, True)
    assert b.fwd_flops(2, 16, 8192, 128) == pytest.approx(0.5498e12, rel=1e-3)
    assert 3.5 * b.fwd_flops(1, 32, 65536, 128) == pytest.approx(123.15e12, rel=1e-3)  # C5 fwd+bwd
    assert b.ROOFLINE_ALGORITHMIC_MB == pytest.approx(1358.9, rel=1e-3)


@pytest.mark.parametrize("n", [2, 4, 8])
def test_local_row_ranges_match_extract_layout(n):
    """bench.parity_check slices the global tensors by these ranges: they must be exactly the rows
    EXTRACT_FUNC_DICT hands to each rank (oracle restatement of extract_local.py:25-49)."""
    b = _bench()
    cfg = dict(b.CONFIG_WORKLOADS[n]); cfg["S"] = 64 * n     # small stand-in with the same grid
    rows = np.arange(cfg["S"], dtype=np.float32).reshape(1, cfg["S"], 1, 1)
    for rank in range(n):
        want = O.EXTRACT[cfg["impl"]](rows, rank, n, cfg["rd"], cfg["ud"])[0, :, 0, 0]
        got = np.concatenate([np.arange(a, e) for a, e in b.local_row_ranges(cfg, rank, n)])
        assert np.array_equal(got, want), (rank, got, want)


@pytest.mark.parametrize("n", [1, 8])
def test_parity_check_finds_a_wrong_row(n):
    """bench.parity_check on host tensors (no aten efficient op there -> None for that leg): an exact shard passes
    the sampled fp64 rows, a shard with one corrupted sampled row (the last owned row is always sampled) does not."""
    b = _bench()
    cfg = dict(b.WORKLOADS[n]); cfg.update(S=32 * n, B=1, Hq=4, Hkv=2, D=16)
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn((1, cfg["S"], h, 16), generator=g) for h in (4, 2, 2))
    full, _ = O.attention_ref(*(t.numpy().astype(np.float64) for t in (q, k, v)), causal=True)
    rank = n - 1
    shard = torch.from_numpy(O.EXTRACT[cfg["impl"]](full, rank, n, cfg["rd"], cfg["ud"])).float()
    op_err, row_err = b.parity_check(cfg, rank, n, shard, q, k, v)
    assert op_err is None and row_err < 1e-5
    bad = shard.clone(); bad[:, -1] += 0.5
    assert b.parity_check(cfg, rank, n, bad, q, k, v)[1] > 0.4
    assert len(b.kernel_source_sha16()) == 16 and b.pmc_traffic() in (None, b.pmc_traffic())


def _probe_worker(rank, ws, ud, rd):
    import torch.distributed as dist
    import yunchang_amd as Y
    import yunchang_amd.hybrid.async_attn_layer as AL
    from yunchang_amd.kernels import set_block_backend
    from oracle_backend import OracleBlockBackend
    b = _bench()
    set_block_backend(OracleBlockBackend())
    Y.set_seq_parallel_pg(ud, rd, rank, ws)
    AL._FILL_ITEMS = 1                       # tiny problem: let the head-group pipeline form
    AL._COMM_OVERRIDE.update(pipeline="1")   # ... also beside a ring (what bench.py's second, overlapped pass runs)
    torch.manual_seed(0)
    B, S, H, D = 1, 32 * ws, 4, 32
    glob = [torch.randn(B, S, H, D).to(torch.bfloat16) for _ in range(3)]
    lq, lk, lv = (Y.EXTRACT_FUNC_DICT["zigzag"](t, rank, world_size=ws, rd=rd, ud=ud) for t in glob)
    attn = Y.LongContextAttention(ring_impl_type="zigzag")
    assert b.exchange_mode(attn, lq, lk, dict(ud=ud, rd=rd, Hq=H, Hkv=H, B=B), ws).startswith(
        "none" if ud == 1 else "one packed q|k|v exchange per head group, 2 group(s), pipelined")
    ref = attn(lq, lk, lv, causal=True)
    dev = torch.device("cpu")
    t = b.timed(lambda: attn(lq, lk, lv, causal=True), 2, ws, dev)
    ov = b.overlap_probe(lambda: attn(lq, lk, lv, causal=True), 2, ws, dev, t)
    again = attn(lq, lk, lv, causal=True)          # the probe must restore comm and the block backend
    assert torch.equal(ref, again)
    return t > 0 and set(ov) >= {"value", "ms_iter", "ms_compute_only", "ms_comm_only"}


@pytest.mark.parametrize("ws,ud,rd", [(2, 1, 2), (4, 2, 2), (4, 1, 4)])
def test_timed_and_overlap_probe_on_gloo(ws, ud, rd):
    """bench.timed / bench.overlap_probe on the N > 1 code path: a ring, and the packed + pipelined exchange beside
    a ring (compute-only swaps the wire for local copies, comm-only skips every kernel; both are restored)."""
    assert all(run_distributed(_probe_worker, ws, ud, rd))


_DEADLINE_SCRIPT = """
import importlib.util, sys, time
spec = importlib.util.spec_from_file_location("bench_mod", sys.argv[1])
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
emit = b._LineOnce({"value": 1.0})
with b._Deadline(float(sys.argv[2]), emit, {"overlap": {"value": None, "error": "deadline"}}):
    time.sleep(float(sys.argv[3]))
emit({"overlap": {"value": 0.5}})
emit()                                  # a second call prints nothing
"""


@pytest.mark.parametrize("deadline,body,expect", [(0.3, 30.0, None), (30.0, 0.05, 0.5)])
def test_probe_deadline_prints_the_line_exactly_once(deadline, body, expect):
    """A probe stuck in a collective must not cost the measurement: at the deadline the line is printed without the
    probe's entry and the process leaves with exit code 0; a probe that finishes prints the line with it.  One line."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", _DEADLINE_SCRIPT, os.path.join(ROOT, "bench.py"), str(deadline), str(body)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["value"] == 1.0 and line["overlap"]["value"] == expect


def _main_worker(rank, ws):
    """bench.main() end to end on host tensors: gloo group of the test harness, the oracle as block backend, tiny
    workloads, the kernel-only probes (which need the device) stubbed."""
    import contextlib
    import io
    import json
    import yunchang_amd.hybrid.async_attn_layer as AL
    from yunchang_amd.kernels import set_block_backend
    from oracle_backend import OracleBlockBackend
    b = _bench()
    set_block_backend(OracleBlockBackend())
    AL._FILL_ITEMS = 1
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank), USP_BENCH_BACKEND="gloo")
    for n, w in b.WORKLOADS.items():
        w.update(B=1, S=64 * n, Hq=4, Hkv=4 if n < 8 else 2, D=32)
    b.kernel_roofline = lambda cfg, dev, traffic=None: {"achieved": 1.0, "stub": True}
    b.reference_kernel = lambda cfg, dev, ours: {"stub": True}
    b.cpu_baseline = lambda cfg: {"stub": True}
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        b.main(["--gpus", str(ws), "--steps", "2", "--warmup", "1"], dev=torch.device("cpu"))
    lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith("{")]
    if rank != 0:
        return lines == []
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    need = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "parity_max_abs_err_vs_fp64_rows"}
    assert need <= set(line), need - set(line)
    assert line["n_gpus"] == ws and line["steps"] == 2 and line["ms_per_step"] > 0, line
    assert line["parity_max_abs_err_vs_fp64_rows"] < 2e-2
    # the line describes itself: environment switches, what the library derived (and from what), per-rank spread
    c = line["config"]
    assert c["env"].get("USP_BENCH_BACKEND") == "gloo" and all(k.startswith(("USP_", "NCCL_", "RCCL_", "HSA_", "HIP_", "GPU_", "TORCH_NCCL_")) for k in c["env"])
    d = c["derived"]
    assert {"link_rate_GBs", "link_rate_source", "kernel_rate_TFs", "kernel_rate_source", "device_cus", "head_group_fill_items",
            "safe_comm", "exchange_relay"} <= set(d), d
    assert d["link_rate_source"].startswith("constant") and d["kernel_rate_source"].startswith("constant")   # gloo: no probe
    assert d["head_group_fill_items"]["source"] == "pinned"          # this test pins AL._FILL_ITEMS = 1
    if c["parallelism"].startswith("ulysses1x"):
        assert "head_groups" not in d
    else:
        assert d["head_groups"] >= 1 and isinstance(d["exchange_link_bound"], bool)
    sp = line["ms_per_step_rank_min_max"]
    assert 0 < sp["min"] <= sp["max"] <= line["ms_per_step"] * 1.0001 + 1e-3, (sp, line["ms_per_step"])
    if ws == 1:
        assert {"roofline", "cpu_baseline", "reference_kernel_on_this_gpu"} <= set(line) and "overlap" not in line
        assert line["roofline"]["stub"] is True and line["scaling"] == "strong" and line["config"]["pass"] == "fwd+bwd"
    else:
        assert set(line["overlap"]) >= {"value", "ms_iter", "ms_compute_only", "ms_comm_only"}
    if ws == 8:       # ulysses 2 x ring 4: two communicators -> the safe mode is measured first, then the library default (overlapped:
                      # pipelined exchange, self-chunk start, row-chunked tails) and the relayed pair exchange on top, each under its deadline
        assert set(line["comm_modes_ms_per_step"]) == {"safe", "overlapped", "relayed"}
        assert line["config"]["comm_mode"].startswith(("safe", "overlapped"))
        assert line["ms_per_step"] == min(line["comm_modes_ms_per_step"].values())
    elif ws == 2:     # ulysses 2 at ring degree 1: round 5's schedule first, the library default (self-chunk start + tails) under a deadline
        assert set(line["comm_modes_ms_per_step"]) == {"plain", "default"}
        assert line["ms_per_step"] == min(line["comm_modes_ms_per_step"].values())
    else:
        assert "comm_modes_ms_per_step" not in line
    return True


@pytest.mark.parametrize("ws", [1, 2, 4, 8])
def test_bench_main_prints_one_contract_line(ws):
    """The driver's contract on every GPU count it launches (N = 1, 2, 4, 8 -> ulysses x ring grids 1x1, 2x1, 1x4,
    2x4 with GQA and a backward): rank 0 prints exactly ONE JSON line carrying the contract keys, the in-bench
    parity of every rank's shard and -- for N > 1 -- the overlap probe; other ranks print nothing."""
    assert all(run_distributed(_main_worker, ws))


def test_profile_is_quoted_only_for_the_kernel_it_was_taken_from(monkeypatch):
    """bench.pmc_traffic(): the committed PMC figures are quoted when the kernel sources are the profiled ones, or --
    sources changed -- when the MACHINE CODE of the profiled kernel inside the shipped library is what was profiled
    (tools/kernel_isa.py: the K-split instantiations were added beside kernels that did not change); otherwise no figure."""
    import sys
    b = _bench()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_isa
    roof, allp, n = kernel_isa.isa_identity(os.path.join(ROOT, "long-context-attention_amd", "libusp_hip.so"))
    assert n == 24 and len(roof) == 16
    newest = os.path.join(ROOT, "profiles", "r06_rocprof_summary.txt")              # what pmc_traffic reads
    t = b.pmc_traffic()
    if not os.path.exists(newest):
        assert t is None
    else:
        lines = open(newest).read().split("\n")
        recorded = [ln.split(":")[1].split()[0] for ln in lines if ln.startswith("roofline_kernel_isa_sha16:")]
        profiled_src = [ln.split(":")[1].strip() for ln in lines if ln.startswith("kernel_src_sha16:")]
        if recorded == [roof] or profiled_src == [b.kernel_source_sha16()]:
            assert t["read_MB"] > 1000 and t["write_MB"] > 100 and t["kernel"] == b.ROOFLINE_KERNEL
        else:
            assert t["read_MB"] is None and "stale" in t
    if t is None:                        # (no profile of this round yet: nothing below to check)
        return
    monkeypatch.setattr(kernel_isa, "isa_identity", lambda lib: ("0" * 16, "0" * 16, 24))      # a different kernel
    monkeypatch.setattr(b, "kernel_source_sha16", lambda: "f" * 16)
    t = b.pmc_traffic()
    assert t["read_MB"] is None and "no figure is claimed" in t["stale"]


def test_sampled_parity_reference_equals_autograd():
    """bench.sampled_parity (the fp64 rows / key columns the 64K entry of the bench line and the full-size GPU tests are
    checked against) on a small GQA problem where the whole answer is affordable: exact attention through autograd in
    fp64 must come out with zero error, a perturbed gradient with exactly its perturbation."""
    b = _bench()
    torch.manual_seed(0)
    B, S, Hq, Hkv, D = 1, 300, 4, 2, 32
    q, k, v, do = (torch.randn(B, S, h, D, dtype=torch.float64) for h in (Hq, Hkv, Hkv, Hq))
    qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
    ke, ve = (t.repeat_interleave(Hq // Hkv, dim=2) for t in (kk, vv))
    s = torch.einsum("bihd,bjhd->bhij", qq, ke) * D ** -0.5
    s = s.masked_fill(~torch.ones(S, S, dtype=torch.bool).tril(), float("-inf"))
    o = torch.einsum("bhij,bjhd->bihd", torch.softmax(s, -1), ve)
    o.backward(do)
    t = dict(q=q, k=k, v=v, do=do, out=o.detach(), lse=torch.logsumexp(s, -1).detach(), dq=qq.grad, dk=kk.grad, dv=vv.grad)
    res = b.sampled_parity(t)
    assert max(res["max_abs_err"].values()) < 1e-9, res
    t["dk"] = t["dk"].clone()
    t["dk"][0, 0, 0, 3] += 0.25                      # key 0 of kv head 0 is always sampled
    res = b.sampled_parity(t)
    assert abs(res["max_abs_err"]["dk"] - 0.25) < 1e-6
    # the same error in comparator units: 0.25 / (atol + rtol |want|) at that element
    want = float(kk.grad[0, 0, 0, 3].abs())
    assert abs(res["max_err_over_tolerance"]["dk"] - 0.25 / (5e-2 + 5e-2 * want)) < 1e-3, res


def test_sampled_parity_and_parity_check_propagate_nan():
    """A NaN in a compared row must reach the printed figure (Python's max(0.0, nan) is 0.0: bench.nanmax), so that
    every `err < tol` gate on the figure fails."""
    b = _bench()
    assert b.nanmax(0.0, float("nan")) != b.nanmax(0.0, float("nan"))          # NaN
    assert b.nanmax(float("nan"), 1.0) != b.nanmax(float("nan"), 1.0)
    assert b.nanmax(0.5, 0.25) == 0.5
    torch.manual_seed(0)
    B, S, Hq, Hkv, D = 1, 300, 2, 1, 32
    q, k, v, do = (torch.randn(B, S, h, D, dtype=torch.float64) for h in (Hq, Hkv, Hkv, Hq))
    z = lambda t: torch.zeros_like(t)
    base = dict(q=q, k=k, v=v, do=do, out=z(q), lse=torch.zeros(B, Hq, S, dtype=torch.float64), dq=z(q), dk=z(k), dv=z(v))
    for name, idx in (("out", (0, S - 1, 0, 0)), ("dq", (0, 0, Hq - 1, 5)), ("dk", (0, S - 1, 0, 1)), ("dv", (0, 0, 0, 0)),
                      ("lse", (0, 0, S - 1))):
        t = dict(base)
        t[name] = base[name].clone()
        t[name][idx] = float("nan")
        res = b.sampled_parity(t)
        e, r = res["max_abs_err"][name], res["max_err_over_tolerance"][name]
        assert e != e and not (e < 1e9), (name, e)
        assert r != r and not (r < 1.0), (name, r)                  # the comparator-unit figure is gated as `ratio < 1`
    # parity_check (the per-rank check of the layer benchmark): a NaN row of the local output
    cfg = dict(S=64, ud=1, rd=1, impl="basic")
    q16, k16, v16 = (torch.randn(1, 64, 2, 16) for _ in range(3))
    out = torch.zeros(1, 64, 2, 16)
    out[0, 63] = float("nan")                                                  # the last row of a range is always sampled
    _, worst_rows = b.parity_check(cfg, 0, 1, out, q16, k16, v16)
    assert worst_rows != worst_rows


def _first_contact_worker(rank, ws, mode):
    """tools/r06/first_contact.py's stages on gloo ranks (CPU tensors, the test oracle as block kernel): the link stage's pair
    exchange and the parity stage of the real layer on BASELINE's grid for that rank count, in the three communicator modes."""
    spec = importlib.util.spec_from_file_location("first_contact", os.path.join(ROOT, "tools", "r06", "first_contact.py"))
    fc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fc)
    os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(ws)
    link = fc.main(["--stage", "link", "--backend", "gloo"])
    par = fc.main(["--stage", "parity", "--mode", mode, "--backend", "gloo"])
    return link, par


@pytest.mark.parametrize("ws,mode", [(2, "default"), (8, "safe"), (8, "default"), (8, "relay")])
def test_first_contact_stages_run_on_gloo(ws, mode):
    """The command a multi-GPU lease starts with (tools/r06/first_contact.sh N) is exercised here stage by stage, so that the
    first thing that can fail on real devices is the devices."""
    for link, par in run_distributed(_first_contact_worker, ws, mode):
        assert link["stage"] == "link" and link["pair_all_to_all_GBs_per_direction_min_over_pairs"] > 0
        assert par["ok"] and par["mode"] == mode and par["grid"] == {2: "ulysses2xring1", 8: "ulysses2xring4"}[ws], par
    sh = open(os.path.join(ROOT, "tools", "r06", "first_contact.sh")).read()
    for needle in ("--stage link", "--stage parity", "USP_SAFE_COMM=1", "bench.py --gpus $N", "first_contact.json"):
        assert needle in sh
