"""CPU tests of the host-side mirror of the reference interface and of the C-ABI library surface
(no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest
import torch

import yunchang_amd as Y
from yunchang_amd import _C
from yunchang_amd.kernels import AttnType, select_flash_attn_impl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "usp_hip.h")).read()
    declared = set(re.findall(r"\b(usp_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"usp_tensor"}
    assert {"usp_flash_fwd", "usp_flash_bwd", "usp_flash_bwd_workspace_bytes", "usp_bwd_delta", "usp_lse_merge", "usp_copy_rows",
            "usp_cast_from_f32", "usp_add_f32", "usp_abi_version", "usp_strerror", "usp_flash_fwd_workspace_bytes"} <= declared
    lib = ctypes.CDLL(_C.lib_path())
    for name in declared:
        assert hasattr(lib, name), f"libusp_hip.so does not export {name}"
    assert set(_C.EXPORTS) == declared
    L = _C.load()
    assert L.usp_abi_version() == _C.ABI_VERSION == 7
    assert L.usp_last_launch_kinds() == 0                        # nothing launched on this thread
    assert b"head_dim" in L.usp_strerror(-2)


def test_argument_validation_without_launch():
    """Bad arguments are rejected before any launch (safe to call without a GPU)."""
    L = _C.load()
    a = _C.UspFwdArgs()
    assert L.usp_flash_fwd(ctypes.byref(a), None) == -1          # null lse
    buf = ctypes.create_string_buffer(4096)
    addr = (ctypes.addressof(buf) + 15) & ~15
    a.lse = addr
    a.dtype, a.B, a.Sq, a.Sk, a.Hq, a.Hkv, a.D = 0, 1, 16, 16, 2, 2, 96
    a.softmax_scale = 0.1
    assert L.usp_flash_fwd(ctypes.byref(a), None) == -2          # head_dim 96 unsupported
    a.D, a.Hq, a.Hkv = 64, 3, 2
    assert L.usp_flash_fwd(ctypes.byref(a), None) == -2          # Hq % Hkv
    a.Hq, a.softmax_scale = 2, 0.0
    assert L.usp_flash_fwd(ctypes.byref(a), None) == -1
    a.dtype, a.softmax_scale = 7, 0.1
    assert L.usp_flash_fwd(ctypes.byref(a), None) == -1
    assert L.usp_copy_rows(addr, addr, 24, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, None) == -2
    assert L.usp_flash_bwd(None, None) == -1


def test_attn_type_surface():
    ref_members = ["aiter", "fa", "fa3", "flashinfer", "torch_math", "torch_flash", "torch_efficient",
                   "torch_cudnn", "sage_auto", "sage_fp16", "sage_fp16_triton", "sage_fp8",
                   "sage_fp8_sm90", "sparse_sage", "npu"]                 # kernels/__init__.py:38-53
    for v in ref_members:
        assert AttnType.from_string(v).value == v
    assert AttnType.from_string("hip") is AttnType.HIP
    with pytest.raises(ValueError):
        AttnType.from_string("torch")                                     # scripts/run_dit.sh:40 case
    for t in (AttnType.HIP, AttnType.FA, AttnType.TORCH_EFFICIENT):
        for stage in ("fwd-only", "bwd-only", "fwd-bwd"):
            assert callable(select_flash_attn_impl(t, stage))
        with pytest.raises(ValueError):
            select_flash_attn_impl(t, "nope")
    with pytest.raises(ValueError):
        select_flash_attn_impl(AttnType.SAGE_FP8, "fwd-only")
    marker = object()
    assert select_flash_attn_impl(AttnType.SAGE_FP8, "fwd-only", attn_processor=marker) is marker


def test_registry_keys_and_exports():
    assert list(Y.EXTRACT_FUNC_DICT) == ["basic", "strip", "zigzag", "basic_pytorch", "basic_flashinfer",
                                         "basic_npu"]                     # extract_local.py:53-60
    from yunchang_amd.hybrid.utils import RING_IMPL_DICT, RING_IMPL_QKVPACKED_DICT
    assert list(RING_IMPL_DICT) == ["basic", "zigzag", "strip", "basic_pytorch", "basic_flashinfer",
                                    "basic_npu"]                          # hybrid/utils.py:14-21
    assert set(RING_IMPL_QKVPACKED_DICT) == {"basic", "zigzag", "strip", "basic_flashinfer"}
    with pytest.raises(NotImplementedError):
        RING_IMPL_DICT["basic_flashinfer"](None, None, None)      # third-party backend: out of scope
    assert RING_IMPL_DICT["strip"] is Y.stripe_flash_attn_func
    for name in ("LongContextAttention", "set_seq_parallel_pg", "EXTRACT_FUNC_DICT", "AttnType",
                 "PROCESS_GROUP", "zigzag_ring_flash_attn_func", "ring_flash_attn_func", "RingComm",
                 "update_out_and_lse", "basic_extract_local", "zigzag_extract_local", "__version__",
                 "UlyssesAttention", "LongContextAttentionQKVPacked", "stripe_flash_attn_func",
                 "stripe_extract_local"):
        assert hasattr(Y, name), name


def test_layer_requires_process_groups():
    Y.PROCESS_GROUP.ULYSSES_PG = None
    Y.PROCESS_GROUP.RING_PG = None
    with pytest.raises(AssertionError, match="set_seq_parallel_pg"):
        Y.LongContextAttention(ring_impl_type="zigzag")
    with pytest.raises(KeyError):
        Y.PROCESS_GROUP.RING_PG = object()
        Y.LongContextAttention(ring_impl_type="stripe")                   # the key is "strip"
    Y.PROCESS_GROUP.RING_PG = None


def test_cpu_tensors_fail_loudly():
    """No CPU fallback: the HIP backend refuses host tensors."""
    from yunchang_amd.kernels import get_block_backend, hip_attn_forward
    assert get_block_backend().name == "hip"
    q = torch.randn(1, 8, 2, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU"):
        hip_attn_forward(q, q, q, causal=True)
    with pytest.raises(NotImplementedError):
        hip_attn_forward(q, q, q, dropout_p=0.1)


def test_grid_matches_reference_layout():
    """globals.py:39-57 replayed by the oracle: ud=2, rd=4, ws=8 -> ulysses {0,1}.., ring {0,2,4,6}.."""
    from oracle import usp_oracle as O
    u, r = O.seq_parallel_groups(2, 4, 8)
    assert u == [[0, 1], [2, 3], [4, 5], [6, 7]] and r == [[0, 2, 4, 6], [1, 3, 5, 7]]
    u, r = O.seq_parallel_groups(2, 2, 8)           # dp = 2
    assert u == [[0, 1], [2, 3], [4, 5], [6, 7]] and r == [[0, 2], [1, 3], [4, 6], [5, 7]]


# ---- ABI: header == ctypes binding == INTEGRATION.md stub == compiled layout -----------------------------------
_CTYPE = {"int32_t": "c_int", "int64_t": "c_long", "float": "c_float", "void*": "c_void_p", "float*": "c_void_p",
          "const float*": "c_void_p", "const int32_t*": "c_void_p", "int32_t*": "c_void_p", "usp_tensor": "T"}


def _header_structs():
    """{struct name: [(C type, field name), ...]} parsed from include/usp_hip.h."""
    import re
    text = open(os.path.join(ROOT, "include", "usp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for body, name in re.findall(r"typedef struct \w+ \{(.*?)\} (\w+);", text, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            ctype, names = decl.rsplit(" ", 1)[0], decl
            m = re.match(r"(const )?(\w+)( ?\*)? (.+)", decl)
            ctype = (m.group(1) or "") + m.group(2) + ("*" if m.group(3) else "")
            for n in m.group(4).split(","):
                fields.append((ctype, n.strip()))
        out[name] = fields
    return out


def _ctypes_fields(cls):
    return [(n, getattr(t, "__name__")) for n, t in cls._fields_]


def test_abi_structs_match_the_header(tmp_path):
    """One field list, four places: include/usp_hip.h, the ctypes binding (_C.py), the stub INTEGRATION.md shows a
    maintainer of the reference, and the layout gcc gives the header (sizeof / offsetof)."""
    import ctypes
    import re
    import subprocess
    from yunchang_amd import _C
    H = _header_structs()
    assert set(H) == {"usp_tensor", "usp_fwd_args", "usp_bwd_args"}
    # (1) the stub in INTEGRATION.md: exec its struct definitions
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    a, b = md.index("i32, i64, f32, vp ="), md.index("assert _L.usp_abi_version()")
    ns = {"ctypes": ctypes}
    exec(md[a:b], ns)
    pairs = {"usp_tensor": (_C.UspTensor, ns["T"]), "usp_fwd_args": (_C.UspFwdArgs, ns["FwdArgs"]),
             "usp_bwd_args": (_C.UspBwdArgs, ns["BwdArgs"])}
    # (2) a compiled probe of the header
    src = tmp_path / "probe.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "usp_hip.h"', 'int main(void) {']
    for sname, fields in H.items():
        lines.append(f'  printf("{sname} %zu\\n", sizeof({sname}));')
        lines += [f'  printf("{sname}.{n} %zu\\n", offsetof({sname}, {n}));' for _, n in fields]
    lines += ['  return 0;', '}']
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    layout = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for sname, fields in H.items():
        for cls in pairs[sname]:
            got = _ctypes_fields(cls)
            assert [n for n, _ in got] == [n for _, n in fields], f"{sname}: field names of {cls.__name__} differ"
            for (n, tname), (ctype, _) in zip(got, fields):
                want = _CTYPE[ctype]
                assert tname == want or (want == "T" and tname in ("T", "UspTensor")), f"{sname}.{n}: {tname} vs {ctype}"
                assert getattr(cls, n).offset == int(layout[f"{sname}.{n}"]), f"{sname}.{n}: offset"
            assert ctypes.sizeof(cls) == int(layout[sname]), f"sizeof({sname}) vs {cls.__name__}"
    assert _C.ABI_VERSION == int(re.search(r"#define USP_ABI_VERSION (\d+)", open(
        os.path.join(ROOT, "include", "usp_hip.h")).read()).group(1))


def test_head_group_cap_is_link_aware():
    """hybrid/async_attn_layer.py: two 256-row items per CU per head-group launch, one where the exchange of a forward
    pass is at least half as long as its attention (BASELINE configs[2] is link-bound, configs[4] is not)."""
    from yunchang_amd.hybrid.async_attn_layer import _groups, _link_bound
    c3 = _link_bound(16, 16, 2, 1, 16384, 128, 2, 1, True)            # 2 GPUs, ulysses 2, MHA, forward
    c5 = _link_bound(32, 4, 2, 1, 16384, 128, 2, 4, True)             # 8 GPUs, ulysses 2 x ring 4, GQA
    assert c3 and not c5
    assert _groups(16, 16, 2, 1, 16384, link_bound=c3)[0] == 2 and _groups(16, 16, 2, 1, 16384)[0] == 1
    assert _groups(32, 4, 2, 1, 16384, link_bound=c5) == (2, 1, 8)
    assert _groups(32, 32, 8, 1, 131072)[0] == 4                      # long sequence: plenty of items, attention-bound
    assert _groups(8, 8, 2, 1, 2048, link_bound=True)[0] == 1         # too small to split at all
    # with the forward K split on (staged), a link-bound forward-only call takes groups of 128 items: four at configs[2]
    assert _groups(16, 16, 2, 1, 16384, link_bound=c3, k_split=True)[0] == 4
    assert _groups(32, 4, 2, 1, 16384, link_bound=c5, k_split=True) == (2, 1, 8)
    assert _groups(16, 16, 1)[0] == 1                                 # no exchange, nothing to hide
    with pytest.raises(AssertionError):
        _groups(6, 6, 4)


def test_item_deal_is_a_balanced_bijection(tmp_path):
    """csrc/usp_item_deal.h (the map the flash kernels use to hand the tiles of a head that spans several XCDs to those
    XCDs) compiled for the host: for every (heads, tiles per head) it must permute the item ids -- a skipped or doubled
    id would be a silently wrong result -- keep every id inside its head, keep every run sorted heaviest first, and,
    where it deals at all, give the eight runs equal weight."""
    import subprocess
    src = tmp_path / "deal.c"
    src.write_text('#define USP_DEAL_FN\n#include "usp_item_deal.h"\n')
    lib = tmp_path / "libdeal.so"
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-I",
                           os.path.join(ROOT, "long-context-attention_amd", "csrc"), str(src), "-o", str(lib)])
    deal = ctypes.CDLL(str(lib)).usp_deal_item
    deal.restype = ctypes.c_int
    dealt_cases = 0
    for heads in range(1, 41):
        for n_inner in (1, 2, 3, 4, 8, 12, 16, 20, 24, 32, 64, 96, 128, 256):
            n = heads * n_inner
            if n % 8:                                   # ItemWalk only splits by XCD when 8 divides the item count
                continue
            items_l = n // 8
            ids = [deal(w, n_inner, items_l) for w in range(n)]
            assert sorted(ids) == list(range(n)), (heads, n_inner)
            if ids == list(range(n)):
                continue
            dealt_cases += 1
            assert heads < 8, (heads, n_inner)          # 8 or more heads keep their contiguous runs
            weights = []
            for x in range(8):
                run = ids[x * items_l:(x + 1) * items_l]
                tiles = [t % n_inner for t in run]
                per_head = {}
                for t in run:
                    per_head.setdefault(t // n_inner, []).append(t % n_inner)
                assert all(v == sorted(v) for v in per_head.values()), (heads, n_inner, x)   # heaviest first per head
                weights.append(sum(n_inner - t for t in tiles))
            regular = n_inner % items_l == 0
            rounds = items_l if regular else n_inner // 8          # tiles a run gets from one head
            if rounds % 2 == 0:                         # whole (up, down) round pairs: exactly equal shares
                assert len(set(weights)) == 1, (heads, n_inner, weights)
            else:                                       # one unpaired round: shares differ by at most its spread
                spread = (n_inner // items_l - 1) if regular else 7 * heads
                assert max(weights) - min(weights) <= spread, (heads, n_inner, weights)
    assert dealt_cases > 30
    # without the XCD split (items_l = all items) nothing is remapped
    assert all(deal(w, 64, 256) == w for w in range(256))
    # round 6: usp_group_item -- the query heads of one KV group side by side inside a run of whole heads: a bijection of every
    # run onto itself, every head's tiles still heaviest first, consecutive ids = the same tile of m heads of ONE KV group, and the
    # alternating passes of an XCD's 32 workgroups still give every workgroup the same weight
    group = ctypes.CDLL(str(lib)).usp_group_item
    group.restype = ctypes.c_int
    grouped_cases = 0
    for heads, G in ((32, 8), (16, 1), (16, 2), (16, 16), (64, 8), (8, 4), (24, 3), (32, 4), (40, 8)):
        for n_inner in (4, 32, 64, 256):
            n = heads * n_inner
            items_l = n // 8
            ids = [group(deal(w, n_inner, items_l), n_inner, items_l, G) for w in range(n)]
            assert sorted(ids) == list(range(n)), (heads, G, n_inner)
            if ids == list(range(n)):
                assert G == 1 or items_l < 2 * n_inner or (items_l // n_inner) % min(items_l // n_inner, G) or G % min(items_l // n_inner, G)
                continue
            grouped_cases += 1
            hr = items_l // n_inner
            m = min(hr, G)
            for x in range(8):
                run = ids[x * items_l:(x + 1) * items_l]
                assert sorted(run) == list(range(x * items_l, (x + 1) * items_l))          # a run stays a run
                per_head = {}
                for t in run:
                    per_head.setdefault(t // n_inner, []).append(t % n_inner)
                assert all(v == sorted(v) for v in per_head.values())                      # heaviest first per head
                for i in range(0, items_l, m):                                             # m consecutive ids: one tile, one KV group
                    blk = run[i:i + m]
                    assert len({t % n_inner for t in blk}) == 1 and len({(t // n_inner) // G for t in blk}) == 1, (heads, G, n_inner, blk)
                if items_l % 64 == 0:                                                       # 32 workgroups, alternating passes
                    wg = [0] * 32
                    for loc, t in enumerate(run):
                        ps, k = divmod(loc, 32)
                        wg[k if ps % 2 == 0 else 31 - k] += n_inner - t % n_inner
                    assert max(wg) - min(wg) <= max(1, n_inner // 8), (heads, G, n_inner, wg)
    assert grouped_cases >= 12


def test_zigzag_fetch_plan_and_wave_matching():
    """Pure schedule logic of the zigzag mesh fetch for every ring degree 3..16, rank, piece count 1..4, one launch per
    source rank or one per query range (batch 1):
    * plan: launches come wave by wave; the score entries per source rank add up to the reference's 2 c^2; every q row
      is emitted exactly once, by the LAST launch that touches it (a later merge would read a stale running output);
    * waves: what rank a posts as sends to b in a wave is what b posts as receives from a (a grouped send/recv call
      that does not pair up hangs RCCL), and no rank posts an empty call."""
    from yunchang_amd.ring.utils import zigzag_wave_steps
    from yunchang_amd.ring.zigzag_ring_flash_attn import zigzag_fetch_plan
    c = 24
    for P in range(3, 17):
        for W in (1, 2, 3, 4):
            cuts = [(i + 1) * c // W - i * c // W for i in range(W)]
            for r in range(P):
                for grouped in (False, True):
                    plan = zigzag_fetch_plan(P, r, W, c, grouped)
                    assert [w for w, *_ in plan] == sorted(w for w, *_ in plan)                  # wave-major
                    if grouped:        # at most two launches per wave: every q row x steps 1..r, q[c:] x steps r+1..P-1
                        assert len(plan) == W * (r >= 1) + 2 * W * (r < P - 1)
                    else:
                        assert len(plan) == W * (P - 1) + W * (P - 1 - r)
                    per_step = {}
                    done = [c if r == 0 else 0, 0]        # rows emitted so far [front, back]; step 0 emits rank 0's front rows
                    for w, lo, hi, all_rows, fe in plan:
                        assert 1 <= lo <= hi < P and (w < W or lo > r) and all_rows == (hi <= r) and (all_rows or lo > r)
                        for step in range(lo, hi + 1):
                            per_step[step] = per_step.get(step, 0) + (2 * c if all_rows else c) * cuts[w % W]
                        if all_rows:
                            assert done == [0, 0], "touches rows that are final"
                            assert fe in (0, c, 2 * c)
                            done[0] += min(fe, c); done[1] += max(fe - c, 0)
                        else:
                            assert done[1] == 0, "touches back rows that are final"
                            assert fe in (0, c)
                            done[1] += fe
                    assert done == [c, c]
                    assert all(n == 2 * c * c for n in per_step.values()) and len(per_step) == P - 1
            # grouped calls pair up: sends of a to b in wave w == receives b expects from a in wave w
            for front in (True, False):
                sends = {(a, (a + s) % P) for a in range(P) for s in zigzag_wave_steps(P, a, front)[0]}
                recvs = {((b - s) % P, b) for b in range(P) for s in zigzag_wave_steps(P, b, front)[1]}
                assert sends == recvs
                assert all(any(a == x or b == x for a, b in sends) for x in range(P))


def test_forward_k_split_workspace_and_policy(monkeypatch):
    """ABI v4's K split of few-item forward launches: the workspace size the C side asks for (host code, no GPU), the
    argument checks that need no launch, and the policy of the Python binding."""
    import ctypes
    L = _C.load()
    a = _C.UspFwdArgs()
    a.B, a.Sq, a.Hq, a.Hkv, a.D, a.Sk = 2, 300, 4, 2, 64, 712
    rows = 2 * 300 * 4
    for n in (2, 3, 8):
        assert L.usp_flash_fwd_workspace_bytes(ctypes.byref(a), n) == n * (rows * 64 + rows) * 4
    assert L.usp_flash_fwd_workspace_bytes(ctypes.byref(a), 1) == 0
    assert L.usp_flash_fwd_workspace_bytes(ctypes.byref(a), 9) == 0
    a.seq_q = 8                                                      # packed batches are not split
    assert L.usp_flash_fwd_workspace_bytes(ctypes.byref(a), 2) == 0
    # the policy (read from USP_FWD_KSPLIT at import; set_fwd_ksplit in-process): only causal launches with fewer than
    # two 256-row items per CU
    monkeypatch.setattr(_C, "_KSPLIT_MODE", _C._parse_ksplit("0"))
    assert _C.fwd_k_splits(1, 16384, 2, True) == 0
    with pytest.warns(UserWarning):
        assert _C._parse_ksplit("fast") == 0                        # a typo must not raise on every launch
    assert _C._parse_ksplit(None) == "auto" and _C._parse_ksplit("1") == 0 and _C._parse_ksplit("99") == 8
    assert _C.set_fwd_ksplit(4) == 0
    assert _C.fwd_k_splits(1, 16384, 2, True) == 4                  # 128 items
    assert _C.fwd_k_splits(1, 16384, 2, False) == 0                 # not causal
    assert _C.fwd_k_splits(1, 16384, 8, True) == 0                  # 512 items: fills the part
    assert _C.fwd_k_splits(2, 8192, 16, True) == 0                  # BASELINE configs[1]
    assert _C.set_fwd_ksplit("auto") == 4
    assert _C.fwd_k_splits(1, 16384, 4, True) == 2 and _C.fwd_k_splits(1, 16384, 2, True) == 4
    assert _C.fwd_k_splits(1, 2048, 2, True) == 0                   # short sequences: nothing to balance


def test_forward_k_split_glue_of_the_binding(monkeypatch):
    """_C.flash_fwd's handling of k_splits without a GPU: the launch is intercepted (a stand-in records the argument
    struct), everything in front of it is the real code -- the policy, the C side's workspace size, the scratch buffer
    (one per stream, reused, grown on demand), the two ABI v4 fields."""
    import ctypes
    real = _C.load()
    seen = []

    class Lib:
        usp_flash_fwd_workspace_bytes = real.usp_flash_fwd_workspace_bytes
        usp_strerror = real.usp_strerror

        @staticmethod
        def usp_flash_fwd(args, stream):
            a = ctypes.cast(args, ctypes.POINTER(_C.UspFwdArgs)).contents
            seen.append((a.k_splits, a.workspace, a.B, a.Sq, a.Hq, a.D, a.causal, a.final_end))
            return 0

    class Stream:
        cuda_stream = 7

    monkeypatch.setattr(_C, "load", lambda: Lib)
    monkeypatch.setattr(_C, "_require_cuda", lambda *t: None)
    monkeypatch.setattr(_C, "_stream", lambda: ctypes.c_void_p(7))
    monkeypatch.setattr(_C.torch.cuda, "current_stream", lambda *a: Stream)
    monkeypatch.setattr(_C, "_FWD_WS", {})
    monkeypatch.setattr(_C, "_KSPLIT_MODE", 0)
    B, S, H, D = 1, 4096, 2, 128
    q = torch.zeros(B, S, H, D, dtype=torch.bfloat16)
    lse = torch.zeros(B, H, S)
    out = torch.zeros_like(q)
    _C.flash_fwd(q, q, q, 0.1, True, lse, out)                       # policy off: no split
    assert seen[-1][:2] == (0, None)
    _C.flash_fwd(q, q, q, 0.1, True, lse, out, k_splits=2)
    n, ws = seen[-1][:2]
    need2 = 2 * (B * S * H * D + B * H * S) * 4
    assert n == 2 and ws and ws % 16 == 0
    (buf,) = _C._FWD_WS.values()
    assert buf.data_ptr() == ws and buf.numel() >= need2
    _C.flash_fwd(q, q, q, 0.1, True, lse, out, k_splits=2)           # reused
    assert seen[-1][1] == ws and len(_C._FWD_WS) == 1
    _C.flash_fwd(q, q, q, 0.1, True, lse, out, k_splits=4)           # grown
    (buf,) = _C._FWD_WS.values()
    assert seen[-1][0] == 4 and buf.numel() >= 2 * need2 and seen[-1][1] == buf.data_ptr()
    monkeypatch.setattr(_C, "_KSPLIT_MODE", "auto")                  # the default policy: 2 heads x 16 tiles -> n = 4
    _C.flash_fwd(q, q, q, 0.1, True, lse, out)
    assert seen[-1][0] == 4
    _C.flash_fwd(q, q, q, 0.1, False, lse, out)                      # not causal -> off
    assert seen[-1][:2] == (0, None)


def test_receive_slot_cache_is_bounded():
    """The persistent K/V receive slots are kept per shape; packed batches change shape every step, so the cache drops
    its least recently used sets instead of growing for ever."""
    from collections import OrderedDict
    from yunchang_amd.ring import utils as U
    cache, made = OrderedDict(), []
    for i in range(U._MAX_SLOT_SETS + 5):
        U._cached_slots(cache, ("shape", i), lambda i=i: made.append(i) or [i])
        U._cached_slots(cache, ("shape", 0), lambda: made.append("again") or ["again"])      # the hot shape stays
    assert len(cache) == U._MAX_SLOT_SETS and ("shape", 0) in cache and "again" not in made
    assert ("shape", 1) not in cache and ("shape", U._MAX_SLOT_SETS + 4) in cache


def test_host_launch_plumbing_without_a_device():
    """On a box without a GPU a well-formed call must get through the whole host side of usp_flash_fwd -- argument
    checks, workgroup shape, the plain and the K-split argument blocks -- and come back with USP_ELAUNCH from the
    launch itself (not crash, not report success).  Skipped where a device exists: the pointers are made up."""
    import ctypes
    if torch.cuda.is_available():
        pytest.skip("a device is present: the made-up pointers would be dereferenced")
    L = _C.load()
    B, S, H, D = 1, 512, 2, 128
    base = 0x7F0000000000
    t = lambda i: _C.UspTensor(base + i * 0x1000000, S * H * D, H * D, D)
    a = _C.UspFwdArgs()
    a.dtype, a.B, a.Sq, a.Sk, a.Hq, a.Hkv, a.D, a.causal = 0, B, S, S, H, H, D, 1
    a.softmax_scale = D ** -0.5
    a.q, a.k, a.v, a.out = t(0), t(1), t(2), t(3)
    a.lse, a.lse_stride_b, a.lse_stride_h = base + 4 * 0x1000000, H * S, S
    a.final_end = S
    assert L.usp_flash_fwd(ctypes.byref(a), None) == -3                      # USP_ELAUNCH
    a.k_splits, a.workspace = 2, base + 5 * 0x1000000
    assert L.usp_flash_fwd(ctypes.byref(a), None) == -3
    a.workspace = base + 5 * 0x1000000 + 4                                   # misaligned scratch: refused before any launch
    assert L.usp_flash_fwd(ctypes.byref(a), None) == -1
    a.k_splits, a.workspace = 9, base + 5 * 0x1000000
    assert L.usp_flash_fwd(ctypes.byref(a), None) == -1
    # ABI v6: kernel-family selectors.  Both at once: invalid.  A forced 64-row call the family does not serve (head dim
    # 64) is refused BEFORE any launch; one it serves -- since round 5 also a K split -- reaches the launch; nothing was
    # launched: kinds == 0.
    a.k_splits, a.workspace = 0, None
    a.flags = _C.USP_FORCE_ROW64 | _C.USP_FORCE_WAVE32
    assert L.usp_flash_fwd(ctypes.byref(a), None) == -1
    a.flags = _C.USP_FORCE_ROW64
    assert L.usp_flash_fwd(ctypes.byref(a), None) == -3
    a.k_splits, a.workspace = 2, base + 5 * 0x1000000
    assert L.usp_flash_fwd(ctypes.byref(a), None) == -3                      # the split instantiation of the 64-row kernel
    a.k_splits, a.workspace, a.D = 0, None, 64
    for i, x in enumerate((a.q, a.k, a.v, a.out)):
        x.stride_b, x.stride_s, x.stride_h = S * H * 64, H * 64, 64
    assert L.usp_flash_fwd(ctypes.byref(a), None) == -2
    a.flags = _C.USP_FORCE_WAVE32
    assert L.usp_flash_fwd(ctypes.byref(a), None) == -3
    assert L.usp_last_launch_kinds() == 0 and _C.last_launch_kinds() == ()
    with pytest.raises(ValueError):
        _C.set_kernel_family("row32")
    assert _C.set_kernel_family("wave32") == "auto" and _C.set_kernel_family("auto") == "wave32"


def test_link_rate_probe_is_inert_without_rccl(monkeypatch):
    """comm/link.py: the rate the head-group sizing uses is the 64 GB/s constant unless measured (RCCL, > 1 rank) or
    pinned (USP_LINK_GBS); on gloo / one rank the probe does nothing (no collective is posted), and the figure feeds
    `_link_bound` -- a fast link turns BASELINE's 2-GPU config from link-bound to attention-bound."""
    from yunchang_amd.comm import link
    from yunchang_amd.hybrid.async_attn_layer import _link_bound
    monkeypatch.delenv("USP_LINK_GBS", raising=False)
    monkeypatch.setattr(link, "_measured", None)
    assert link.probe_link_rate(0, 1) is None and link.probe_link_rate(0, 8) is None and not link.measured()
    assert link.link_bytes_per_s() == 64e9
    assert _link_bound(16, 16, 2, 1, 16384, 128, 2, 1, True)
    monkeypatch.setenv("USP_LINK_GBS", "400")
    assert link.link_bytes_per_s() == 400e9 and not _link_bound(16, 16, 2, 1, 16384, 128, 2, 1, True)
    monkeypatch.setenv("USP_LINK_GBS", "fast")                   # a typo falls back to the constant
    assert link.link_bytes_per_s() == 64e9
    monkeypatch.delenv("USP_LINK_GBS")
    monkeypatch.setattr(link, "_measured", 90e9)
    assert link.link_bytes_per_s() == 90e9


def test_backward_cut_workspace_and_policy(monkeypatch):
    """ABI v5's cuts of few-item backward launches: the workspace size the C side asks for (host code, no GPU), the
    argument checks that need no launch, and the policy of the Python binding."""
    import ctypes
    L = _C.load()
    a = _C.UspBwdArgs()
    a.B, a.Sq, a.Sk, a.Hq, a.Hkv, a.D = 2, 300, 712, 4, 2, 64
    kv_slab, q_slab = 2 * 712 * 2 * 64 * 4, 2 * 300 * 4 * 64 * 4
    need = lambda: L.usp_flash_bwd_workspace_bytes(ctypes.byref(a))
    assert need() == 2 * 2 * kv_slab                                 # GQA head split alone: G = 2 slabs for dK and for dV
    a.dkdv_splits = 3
    assert need() == 2 * 6 * kv_slab
    a.dq_splits = 4
    assert need() == 2 * 6 * kv_slab + 4 * q_slab
    a.dkdv_splits, a.Hq = 1, 2                                       # MHA, no cut of the dK/dV launch: only dQ partials
    assert need() == 4 * (2 * 300 * 2 * 64 * 4)
    a.dq_splits = 0
    assert need() == 0
    a.dq_splits, a.seq_q = 4, 8                                      # packed batches are not cut
    assert need() == 0
    # the policy
    monkeypatch.setattr(_C, "_BWD_SPLIT_MODE", "auto")
    assert _C.bwd_splits(1, 16384, 16384, 2, True) == (4, 2) and _C.bwd_splits(1, 16384, 16384, 4, True) == (2, 0)
    assert _C.bwd_splits(1, 16384, 16384, 8, True) == (0, 0) and _C.bwd_splits(2, 8192, 8192, 16, True) == (0, 0)
    assert _C.bwd_splits(1, 8192, 16384, 8, False) == (0, 0)        # a ring step of BASELINE's 8-GPU rank: 256 / 1024 items
    assert _C.bwd_splits(1, 2048, 2048, 2, True) == (0, 0)          # short: nothing to balance
    assert _C.set_bwd_split(0) == "auto" and _C.bwd_splits(1, 16384, 16384, 2, True) == (0, 0)
    assert _C.set_bwd_split("auto") == 0


def test_relay_plan_is_a_partition_and_an_involution():
    """comm/relay_exchange.py as pure functions: on every grid it applies to, a rank's peer's peer is the rank itself, the
    helpers are the rest of its sequence-parallel block, the stripes a sender cuts its chunk into cover every row exactly
    once, and sender and receiver agree on where each stripe lies."""
    from yunchang_amd.comm import relay_exchange as R
    for ud, rd, ws, low in [(2, 2, 4, True), (2, 4, 8, True), (2, 4, 8, False), (2, 3, 6, True), (2, 2, 8, True), (2, 8, 16, False)]:
        grid = (ud, rd, ws, low)
        sp = ud * rd
        for rank in range(ws):
            peer, helpers = R.pair_and_helpers(rank, grid)
            assert R.pair_and_helpers(peer, grid)[0] == rank and peer != rank
            base = (rank // sp) * sp
            assert sorted(helpers + [rank, peer]) == list(range(base, base + sp))
            assert R.pair_and_helpers(peer, grid)[1] == helpers             # a pair shares its helpers
            for rows in (8, 37, 1024, 8192):
                d, r = R.stripe_rows(rows, len(helpers))
                assert d + len(helpers) * r == rows and d >= r >= 0
                off = R._offsets(rank, peer, helpers, d, r)
                spans = sorted((off[w], off[w] + (d if w == peer else r)) for w in helpers + [peer])
                assert spans[0][0] == 0 and spans[-1][1] == rows and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert R.pair_and_helpers(0, (4, 2, 8, True)) is None and R.pair_and_helpers(0, (2, 1, 2, True)) is None
    assert R.pair_and_helpers(0, None) is None or R.GRID is not None
