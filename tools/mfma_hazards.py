"""Static check of the software-visible MFMA hazards in a hipcc -save-temps .s file (DEV TOOL, no GPU needed).

    python tools/mfma_hazards.py file.s <kernel-name-substring>

The one-wave-per-SIMD kernels issue their MFMAs from inline asm, and hipcc pads nothing around an asm statement.  Checked
per straight-line instruction stream (labels and branches reset nothing: the stream is checked in layout order, which
over-approximates):
  A  VALU / DS-return write of a VGPR  ->  MFMA reading it as SrcA/B/C within < 2 instructions   (needs 2 wait states)
  B  MFMA writing D  ->  any non-MFMA instruction reading or writing a register of D within < 12 wait states
     (8-pass 32x32x16: 12 states; an s_nop N counts N + 1, an intervening MFMA 8 -- the matrix pipe accepts the next
     32x32x16 MFMA one 8-pass slot after the previous one, so two MFMAs are never less than 8 states apart --, every
     other instruction 1; an MFMA that takes D whole as its SrcC and writes it back is the exempt accumulate chain)
  C  v_accvgpr_write  ->  MFMA reading that AGPR within < 3 instructions
The scan behind an MFMA stops at an unconditional branch (layout order is not execution order there; the target of a
conditional or unconditional branch is NOT followed -- loops re-enter code that has been checked from its own MFMAs).
Prints every violation with its line number in the .s file."""
import re
import sys


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


def check(path, pat, out=print):
    """Number of potential hazards of the kernel whose mangled name contains `pat`; every violation goes to `out`."""
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^(_Z\S*):", l) and pat in l)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    ins = []
    for n in range(start + 1, end):
        s = lines[n].split(";")[0].strip()
        if not s or s.startswith(".") or s.endswith(":"):
            continue
        op, _, rest = s.partition(" ")
        ops = [x.strip() for x in rest.split(",")] if rest else []
        ins.append((n + 1, op, ops))
    bad = 0
    for i, (ln, op, ops) in enumerate(ins):
        if not op.startswith("v_mfma"):
            continue
        d, a, b, c = regs(ops[0]), regs(ops[1]), regs(ops[2]), regs(ops[3]) if len(ops) > 3 else set()
        # A / C: producers just in front
        for k in (1, 2):
            if i - k < 0:
                break
            pl, pop, pops = ins[i - k]
            if pop.startswith(("s_", "v_mfma", "buffer_", "global_", "ds_write", "scratch_store")) or not pops:
                if pop.startswith("s_nop"):
                    break                                   # a nop in between: enough states (s_nop >= 1 emitted as s_nop 1+)
                continue
            w = regs(pops[0])
            if pop.startswith(("ds_read", "buffer_load", "global_load", "scratch_load")):
                continue                                    # returned data is guarded by s_waitcnt, not by wait states
            if w & (a | b | (c - d)):
                out(f"A: line {pl}: {pop} {pops[0]} feeds MFMA at line {ln} ({k} instruction(s) earlier)")
                bad += 1
        # B: consumers of D behind
        states = 0
        for k in range(1, 40):
            if i + k >= len(ins) or states >= int(__import__("os").environ.get("HZ_STATES", "12")):
                break
            nl, nop_, nops = ins[i + k]
            if nop_ in ("s_branch", "s_endpgm", "s_setpc_b64"):
                break                                       # what follows in layout order is not what executes next
            if nop_.startswith("v_mfma"):
                nd, nc = regs(nops[0]), regs(nops[3]) if len(nops) > 3 else set()
                touched = set()
                for t in nops[1:3]:
                    touched |= regs(t)
                if touched & d:
                    out(f"B: line {nl}: MFMA reads D of MFMA at line {ln} as SrcA/B after {states} states")
                    bad += 1
                if (nc & d or nd & d) and not (nc == d and nd == d):
                    out(f"B: line {nl}: MFMA overlaps D of MFMA at line {ln} partially after {states} states")
                    bad += 1
                states += 8
                continue
            touched = set()
            for t in nops:
                touched |= regs(t.split(" ")[0])
            if touched & d and not nop_.startswith("s_"):
                out(f"B: line {nl}: {nop_} {' '.join(nops)[:50]} touches D of MFMA at line {ln} after {states} states")
                bad += 1
            m = re.match(r"s_nop", nop_)
            states += (int(nops[0]) + 1) if m else 1
    return bad


def main():
    bad = check(sys.argv[1], sys.argv[2])
    print(f"{bad} potential hazard(s)")


if __name__ == "__main__":
    main()
