"""Static check of the software-visible MFMA hazards of the one-wave-per-SIMD kernels (DEV / BUILD TOOL, no GPU needed).

    python tools/mfma_hazards.py file.s <kernel-name-substring>      a hipcc -S file
    python tools/mfma_hazards.py --lib [libusp_hip.so]               the SHIPPED library (llvm-objdump disassembly): every
                                                                     kernel that holds an MFMA -- what build() runs

The one-wave-per-SIMD kernels issue their MFMAs from inline asm, and hipcc pads nothing around an asm statement.  Checked
over the CONTROL-FLOW GRAPH of a function (round 5: branch targets are followed, forwards and backwards -- an MFMA at the
end of a loop body is checked against the loop head's first instructions, and the producers in front of an MFMA at a loop
head include the loop's last instructions; round 4 scanned in layout order only):
  A  VALU write of a VGPR  ->  MFMA reading it as SrcA/B/C within < 2 instructions   (needs 2 wait states)
  B  MFMA writing D  ->  any non-MFMA instruction reading or writing a register of D within < 12 wait states
     (8-pass 32x32x16: 12 states; an s_nop N counts N + 1, an intervening MFMA 8 -- the matrix pipe accepts the next
     32x32x16 MFMA one 8-pass slot after the previous one, so two MFMAs are never less than 8 states apart --, every
     other instruction 1; an MFMA that takes D whole as its SrcC and writes it back is the exempt accumulate chain)
  C  v_accvgpr_write  ->  MFMA reading that AGPR within < 3 instructions (covered by A's scan: same producer rule)
Every path is followed: both sides of a conditional branch, the target of an unconditional one; a path ends at s_endpgm /
s_setpc_b64 or when 12 states have passed.  Prints every violation with its position (line of the .s file, or address)."""
import os
import re
import subprocess
import sys

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
NEED_STATES = int(os.environ.get("HZ_STATES", "12"))


def regs(tok):
    """'v[64:79]' / 'v12' / 'a[0:3]' -> set of ('v', n)"""
    tok = tok.strip().rstrip(",")
    m = re.match(r"^([va])\[(\d+):(\d+)\]$", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"^([va])(\d+)$", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    return set()


def kernels_in(path):
    """Mangled names of the kernels (functions that contain an MFMA) of a .s file."""
    names, cur, has = [], None, False
    for l in open(path).read().split("\n"):
        m = re.match(r"^(_Z\S*):", l)
        if m:
            cur, has = m.group(1), False
        elif l.startswith(".Lfunc_end"):
            if cur and has:
                names.append(cur)
            cur = None
        elif cur and "v_mfma" in l:
            has = True
    return names


# ---- front ends: a function -> [(position, mnemonic, operands, branch target index | None)] ------------------------------
def _parse_s(path, pat):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^(_Z\S*):", l) and pat in l)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    raw, labels = [], {}
    for n in range(start + 1, end):
        s = lines[n].split(";")[0].strip()
        if not s:
            continue
        m = re.match(r"^(\.?[\w$.]+):$", s)
        if m:
            labels[m.group(1)] = len(raw)                   # the label names the NEXT instruction
            continue
        if s.startswith("."):
            continue
        op, _, rest = s.partition(" ")
        raw.append((n + 1, op, [x.strip() for x in rest.split(",")] if rest else []))
    ins = []
    for pos, op, ops in raw:
        tgt = labels.get(ops[0]) if op.startswith(("s_branch", "s_cbranch")) and ops else None
        ins.append((pos, op, ops, tgt))
    return ins


def disassemble_library(lib):
    """{demangled kernel name: instruction list} of every function with an MFMA in the code objects bundled in `lib`."""
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix="usp_hz_")
    out = {}
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", local], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "--demangle", os.path.join(tmp, f)], check=True,
                                 capture_output=True, text=True).stdout
            cur, raw = None, []

            def close():
                if cur is not None and any(op.startswith("v_mfma") for _, op, _, _ in raw):
                    by_addr = {a: i for i, (a, _, _, _) in enumerate(raw)}
                    out[cur] = [(f"{a:#x}", op, ops, by_addr.get(t) if t is not None else None) for a, op, ops, t in raw]
            for line in txt.split("\n"):
                m = re.match(r"^([0-9a-f]+) <(.*)>:$", line)
                if m:
                    close()
                    cur, raw, base = m.group(2), [], int(m.group(1), 16)
                    continue
                if cur is None or not line.startswith(("\t", " ")):
                    continue
                code, _, comment = line.partition("//")
                s = re.sub(r"\s+", " ", code.strip())
                am = re.match(r"\s*([0-9A-Fa-f]+):", comment)
                if not s or not am:
                    continue
                op, _, rest = s.partition(" ")
                tgt = None
                if op.startswith(("s_branch", "s_cbranch")):
                    tm = re.search(r"\+0x([0-9a-f]+)>\s*$", comment)
                    tgt = base + int(tm.group(1), 16) if tm else (base if comment.rstrip().endswith(">") else None)
                raw.append((int(am.group(1), 16), op, [x.strip() for x in rest.split(",")] if rest else [], tgt))
            close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


# ---- the check over the control-flow graph ------------------------------------------------------------------------------------
_END = ("s_endpgm", "s_setpc_b64", "s_swappc_b64")


def _successors(ins, i):
    _pos, op, _ops, tgt = ins[i]
    if op in _END:
        return []
    if op == "s_branch":
        return [tgt] if tgt is not None else []
    nxt = [i + 1] if i + 1 < len(ins) else []
    if op.startswith("s_cbranch") and tgt is not None:
        nxt.append(tgt)
    return nxt


def _is_producer(op):
    """A VALU-class write whose result an MFMA may not read for two wait states (loads return under s_waitcnt)."""
    return not op.startswith(("s_", "v_mfma", "buffer_", "global_", "flat_", "ds_", "scratch_"))


def check_stream(ins, out=print, where=""):
    preds = [[] for _ in ins]
    for i in range(len(ins)):
        for j in _successors(ins, i):
            preds[j].append(i)
    bad = 0
    for i, (pos, op, ops, _t) in enumerate(ins):
        if not op.startswith("v_mfma"):
            continue
        d, a, b = regs(ops[0]), regs(ops[1]), regs(ops[2])
        c = regs(ops[3]) if len(ops) > 3 else set()
        # A / C: producers one or two instructions in front, along every way into this MFMA
        front = [(p, 1) for p in preds[i]]
        seen = set()
        while front:
            p, k = front.pop()
            if (p, k) in seen:
                continue
            seen.add((p, k))
            ppos, pop, pops, _ = ins[p]
            if pop.startswith("s_nop"):
                continue                                     # enough states on this way in
            if pops and _is_producer(pop) and regs(pops[0]) & (a | b | (c - d)):
                out(f"A: {where}{ppos}: {pop} {pops[0]} feeds MFMA at {pos} ({k} instruction(s) earlier)")
                bad += 1
            if k < 2:
                front += [(q, k + 1) for q in preds[p]]
        # B: consumers of D behind, along every way out, until NEED_STATES wait states have passed
        best = {}
        work = [(j, 0) for j in _successors(ins, i)]
        while work:
            j, states = work.pop()
            if states >= NEED_STATES or best.get(j, NEED_STATES + 1) <= states:
                continue
            best[j] = states
            npos, nop_, nops, _ = ins[j]
            if nop_.startswith("v_mfma"):
                nd, nc = regs(nops[0]), regs(nops[3]) if len(nops) > 3 else set()
                touched = set()
                for t in nops[1:3]:
                    touched |= regs(t)
                if touched & d:
                    out(f"B: {where}{npos}: MFMA reads D of MFMA at {pos} as SrcA/B after {states} states")
                    bad += 1
                if (nc & d or nd & d) and not (nc == d and nd == d):
                    out(f"B: {where}{npos}: MFMA overlaps D of MFMA at {pos} partially after {states} states")
                    bad += 1
                step = 8
            else:
                touched = set()
                for t in nops:
                    touched |= regs(t.split(" ")[0])
                if touched & d and not nop_.startswith("s_"):
                    out(f"B: {where}{npos}: {nop_} {' '.join(nops)[:50]} touches D of MFMA at {pos} after {states} states")
                    bad += 1
                step = (int(nops[0], 0) + 1) if nop_.startswith("s_nop") and nops else 1
            for nxt in _successors(ins, j):
                work.append((nxt, states + step))
    return bad


def check(path, pat, out=print):
    """Number of potential hazards of the kernel of a .s file whose mangled name contains `pat`."""
    return check_stream(_parse_s(path, pat), out, "line ")


def check_library(lib, out=print, only=("64_kernel",)):
    """(kernels checked, potential hazards) over the kernels of the shipped library whose name contains one of `only`
    (default: the three one-wave-per-SIMD families, whose MFMAs are inline asm; the builtin MFMAs of the 8-wave kernels
    are padded by hipcc itself)."""
    n = bad = 0
    for name, ins in sorted(disassemble_library(lib).items()):
        if only and not any(o in name for o in only):
            continue
        found = []
        k = check_stream(ins, found.append, "")
        n += 1
        bad += k
        for f in found[:10]:
            out(f"{name[:80]}: {f}")
    return n, bad


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--lib":
        lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                 "long-context-attention_amd", "libusp_hip.so")
        n, bad = check_library(lib)
        print(f"{n} kernels checked in {lib}: {bad} potential hazard(s)")
        sys.exit(1 if bad or n == 0 else 0)
    bad = check(sys.argv[1], sys.argv[2])
    print(f"{bad} potential hazard(s)")


if __name__ == "__main__":
    main()
