"""Instruction mix of the loops of a kernel inside libusp_hip.so (DEV TOOL, no GPU needed).

    python tools/loop_stats.py "dkdv_kernel<128, 0, true>" [lib.so] [--dump out.s]

A wave issues in order, about one instruction per 4 cycles, and a v_mfma_f32_32x32x16_bf16 occupies its SIMD's matrix
pipe for 32 cycles: with two waves per SIMD a loop that carries much more than ~8 instructions per MFMA is bound by
instruction issue, not by the pipe.  For every backward branch (= loop) that contains MFMAs this prints the instruction
count, the per-MFMA ratio and what the non-MFMA instructions are (v_readlane = SGPR-spill reloads)."""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_instructions(lib, pattern):
    tmp = tempfile.mkdtemp(prefix="usp_loops_")
    try:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))
        subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([OBJDUMP, "-d", "--demangle", os.path.join(tmp, f)], capture_output=True, text=True).stdout
            lines = txt.split("\n")
            for i, l in enumerate(lines):
                m = re.match(r"^[0-9a-f]+ <(.*)>:$", l)
                if m and pattern in m.group(1):
                    end = next((j for j in range(i + 1, len(lines)) if re.match(r"^[0-9a-f]+ <", lines[j])), len(lines))
                    ins = []
                    for x in lines[i + 1:end]:
                        mm = re.match(r"\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", x)
                        if mm:
                            ins.append((int(mm.group(2), 16), mm.group(1)))
                    return m.group(1), ins
        raise SystemExit(f"no kernel matching {pattern!r} in {lib}")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def loops(ins):
    a2i = {a: i for i, (a, _) in enumerate(ins)}
    out = []
    for i, (a, t) in enumerate(ins):
        if t.startswith(("s_cbranch", "s_branch")):
            off = int(t.split()[-1])
            off = off - 65536 if off >= 32768 else off
            tgt = a + 4 + off * 4
            if tgt < a and tgt in a2i:
                j = a2i[tgt]
                c = collections.Counter(x.split()[0] for _, x in ins[j:i + 1])
                mf = sum(v for k, v in c.items() if "mfma" in k)
                if mf:
                    out.append(dict(first=j, last=i, n=i - j + 1, mfma=mf, readlane=c["v_readlane_b32"],
                                    salu=sum(v for k, v in c.items() if k.startswith("s_") and not k.startswith(("s_waitcnt", "s_nop"))),
                                    valu=sum(v for k, v in c.items() if k.startswith("v_") and "mfma" not in k),
                                    ds=sum(v for k, v in c.items() if k.startswith("ds_")), waitcnt=c["s_waitcnt"], nop=c["s_nop"]))
    return out


if __name__ == "__main__":
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    args = [a for a in sys.argv[1:] if not a.startswith("--") and a != dump]
    lib = args[1] if len(args) > 1 else os.path.join(ROOT, "long-context-attention_amd", "libusp_hip.so")
    name, ins = kernel_instructions(lib, args[0])
    print(f"{name}: {len(ins)} instructions")
    for l in loops(ins):
        print("  loop [{first}, {last}]: {n} instructions, {mfma} MFMA = {r:.1f} per MFMA; SALU {salu}, VALU {valu} "
              "(v_readlane {readlane}), DS {ds}, s_waitcnt {waitcnt}, s_nop {nop}".format(r=l["n"] / l["mfma"], **l))
    if dump:
        with open(dump, "w") as f:
            f.write("\n".join(f"{i} {t}" for i, (_, t) in enumerate(ins)))
