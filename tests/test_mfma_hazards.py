"""Build-time gate for the one-wave-per-SIMD kernels (usp_flash_fwd64.hip, usp_flash_bwd64.hip, usp_flash_bwd_dq64.hip):
their MFMAs are inline asm, and hipcc pads NO hazard around an asm statement -- whether the wait states the ISA asks for
are there depends on what the register allocator and the scheduler did with the code between the statements.  The emitted
instruction stream of every instantiation is checked statically (tools/mfma_hazards.py: VALU / accvgpr write -> MFMA
operand, MFMA result -> any other reader or writer inside 12 wait states).  No GPU: hipcc cross-compiles gfx950."""
import os
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "long-context-attention_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
SOURCES = ["usp_flash_fwd64.hip", "usp_flash_bwd64.hip", "usp_flash_bwd_dq64.hip"]

sys.path.insert(0, os.path.join(ROOT, "tools"))


def _emit(src, out_dir):
    out = os.path.join(out_dir, src.replace(".hip", ".s"))
    # the flags of csrc/Makefile that shape device code
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", f"-I{os.path.join(ROOT, 'include')}",
           "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_no_software_visible_mfma_hazard_in_the_64_row_kernels():
    import mfma_hazards as H
    tmp = tempfile.mkdtemp(prefix="usp_hz_")
    try:
        with ThreadPoolExecutor(len(SOURCES)) as ex:
            files = list(ex.map(lambda s: _emit(s, tmp), SOURCES))
        seen = 0
        for f in files:
            names = H.kernels_in(f)
            assert names, f"no MFMA kernel found in {f}"
            for n in names:
                found = []
                bad = H.check(f, n, out=found.append)
                assert bad == 0, f"{os.path.basename(f)} {n}: {bad} potential hazard(s)\n" + "\n".join(found[:10])
                seen += 1
        assert seen >= 12         # forward, dK/dV and dQ: bf16 / fp16 x causal / full
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_the_checker_sees_a_planted_hazard(tmp_path):
    """Mutation: an MFMA result read two instructions later, and a packed operand written right in front of its MFMA."""
    import mfma_hazards as H
    s = tmp_path / "k.s"
    s.write_text("\n".join([
        "_Z1kv:",
        "\tv_cvt_pk_bf16_f32 v40, v1, v2",
        "\tv_mfma_f32_32x32x16_bf16 v[0:15], v[40:43], v[44:47], v[0:15]",
        "\tv_add_f32_e32 v50, v51, v52",
        "\tv_mul_f32_e32 v60, v3, v61",
        "\ts_endpgm",
        ".Lfunc_end0:", ""]))
    found = []
    assert H.check(str(s), "_Z1kv", out=found.append) == 2, found
    assert any(x.startswith("A:") for x in found) and any(x.startswith("B:") for x in found)
    clean = tmp_path / "c.s"
    clean.write_text("\n".join([
        "_Z1cv:",
        "\tv_cvt_pk_bf16_f32 v40, v1, v2",
        "\ts_nop 1",
        "\tv_mfma_f32_32x32x16_bf16 v[0:15], v[40:43], v[44:47], v[0:15]",
        "\ts_nop 15",
        "\tv_mul_f32_e32 v60, v3, v61",
        "\ts_endpgm",
        ".Lfunc_end0:", ""]))
    assert H.check(str(clean), "_Z1cv", out=found.append) == 0


def test_the_checker_follows_loop_back_edges(tmp_path):
    """Round 4's scan ran in layout order and never followed a branch: an MFMA at the END of a loop body whose result the
    loop HEAD reads was invisible to it, and so was a producer at the loop's end feeding an MFMA at
    its head.  Both planted here; the same loop with the wait states in place is clean."""
    import mfma_hazards as H
    s = tmp_path / "loop.s"
    s.write_text("\n".join([
        "_Z1lv:",
        ".LBB0_1:",
        "\tv_add_f32_e32 v50, v0, v52",                                      # reads D of the MFMA at the loop's end
        "\ts_nop 7",
        "\ts_nop 7",
        "\tv_mfma_f32_32x32x16_bf16 v[0:15], v[40:43], v[44:47], v[0:15]",
        "\ts_cbranch_scc1 .LBB0_1",
        "\ts_nop 15",
        "\ts_endpgm",
        ".Lfunc_end0:", ""]))
    found = []
    assert H.check(str(s), "_Z1lv", out=found.append) == 1 and found[0].startswith("B:"), found
    s2 = tmp_path / "loop2.s"
    s2.write_text("\n".join([
        "_Z1mv:",
        ".LBB0_1:",
        "\tv_mfma_f32_32x32x16_bf16 v[0:15], v[40:43], v[44:47], v[0:15]",   # its A operand is written at the loop's end
        "\ts_nop 15",
        "\tv_cvt_pk_bf16_f32 v40, v1, v2",
        "\ts_cbranch_scc1 .LBB0_1",
        "\ts_endpgm",
        ".Lfunc_end0:", ""]))
    found = []
    assert H.check(str(s2), "_Z1mv", out=found.append) == 1 and found[0].startswith("A:"), found
    ok = tmp_path / "ok.s"
    ok.write_text("\n".join([
        "_Z1ov:",
        ".LBB0_1:",
        "\tv_mfma_f32_32x32x16_bf16 v[0:15], v[40:43], v[44:47], v[0:15]",
        "\ts_nop 15",
        "\tv_cvt_pk_bf16_f32 v40, v1, v2",
        "\ts_nop 1",
        "\ts_cbranch_scc1 .LBB0_1",
        "\ts_nop 15",
        "\tv_add_f32_e32 v50, v0, v52",
        "\ts_endpgm",
        ".Lfunc_end0:", ""]))
    assert H.check(str(ok), "_Z1ov", out=found.append) == 0


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="no llvm-objdump")
def test_no_hazard_in_the_shipped_library():
    """The same check on the DISASSEMBLY of libusp_hip.so as it ships (what __graft_entry__.build() runs): a rebuild with
    another hipcc is guarded by the library it produced, not by a fresh -S compile next to it."""
    import mfma_hazards as H
    lib = os.path.join(ROOT, "long-context-attention_amd", "libusp_hip.so")
    if not os.path.exists(lib):
        pytest.skip("libusp_hip.so not built")
    found = []
    n, bad = H.check_library(lib, out=found.append)
    assert n >= 12 and bad == 0, found[:10]
