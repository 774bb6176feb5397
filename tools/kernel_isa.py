"""Identity of a kernel's MACHINE CODE inside libusp_hip.so (DEV / bench helper, no GPU needed).

    python tools/kernel_isa.py [path/to/libusp_hip.so]      -> sha16 of the roofline kernel + of all plain forward kernels

A profile (profiles/*_rocprof_summary.txt) belongs to the kernel sources it was taken from; bench.py refuses to quote
its PMC figures when the sources have changed.  Sources can change without the profiled kernel changing (round 2 added
the K-split instantiations beside the plain forward kernels): what stays comparable then is the machine code itself.
This extracts the gfx950 code objects from the shared library (llvm-objdump --offloading), disassembles them, and hashes
the instruction stream of one kernel -- mnemonics and operands, no addresses, no symbol names.  The one thing masked is
the PC-relative literal of a call sequence (s_getpc_b64 + s_add_u32/s_addc_u32 <distance>): the distance to the two
non-inlined queue functions depends on what else is linked into the code object, not on the kernel."""
import hashlib
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
ROOFLINE_KERNEL = r"flash_bwd_dkdv64_kernel<0, true>"  # the 4 x 64-key dK/dV kernel, bf16, causal: the dominant kernel of the N=1 step


def _disassemble(lib):
    """{demangled kernel name: [instruction lines]} of every code object bundled in `lib`."""
    tmp = tempfile.mkdtemp(prefix="usp_isa_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", local], cwd=tmp, check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        kernels = {}
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "--demangle", os.path.join(tmp, f)], check=True,
                                 capture_output=True, text=True).stdout
            cur, pcrel = None, 0
            for line in txt.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
                if m:
                    cur = m.group(1)
                    kernels[cur] = []
                    continue
                if cur is None or not line.startswith(("\t", " ")):
                    continue
                ins = re.sub(r"\s+", " ", line.split("//")[0].strip())
                if not ins:
                    continue
                if ins.startswith("s_getpc_b64"):
                    pcrel = 4                                   # the literals of the next few scalar adds are distances
                elif pcrel > 0:
                    pcrel -= 1
                    if ins.startswith(("s_add_u32", "s_addc_u32")):
                        ins = re.sub(r"0x[0-9a-f]+$", "<pcrel>", ins)
                kernels[cur].append(ins)
        return kernels
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _sha16(lines):
    return hashlib.sha256("\n".join(lines).encode()).hexdigest()[:16]


def isa_identity(lib):
    """(sha16 of the roofline kernel's instruction stream, sha16 over all plain 8 / 4-wave forward kernels, their number)."""
    ks = _disassemble(lib)
    plain = {n: b for n, b in ks.items() if re.search(r"usp::flash_fwd_kernel<\d+, \d, (true|false), \d(, false)?>", n)}
    roof = [b for n, b in ks.items() if re.search(ROOFLINE_KERNEL, n)]
    if len(roof) != 1:
        raise RuntimeError(f"roofline kernel not found (or ambiguous) in {lib}: {len(roof)} matches")
    allp = []
    for n in sorted(plain, key=lambda x: re.sub(r", false>", ">", x)):
        allp += plain[n]
    return _sha16(roof[0]), _sha16(allp), len(plain)


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             "long-context-attention_amd", "libusp_hip.so")
    r, a, n = isa_identity(lib)
    print(f"roofline_kernel_isa_sha16: {r}")
    print(f"plain_forward_kernels_isa_sha16: {a}  ({n} kernels)")
    if "--all" in sys.argv:                      # every flash kernel of the library, one line each (refactoring proofs)
        for name, body in sorted(_disassemble(lib).items()):
            if "flash_" in name and "kernel" in name:
                print(f"{_sha16(body)}  {len(body):6d}  {name[:110]}")
