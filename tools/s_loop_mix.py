"""Instruction mix of the MFMA loops in a hipcc -save-temps .s file (DEV TOOL, no GPU needed).

    python tools/s_loop_mix.py file.s <kernel-name-substring> [--dump]

For every loop (backward branch) that contains MFMAs: instruction count by class, per-MFMA ratios, and the
instructions that should not be in a pipelined loop (scratch traffic, v_accvgpr copies, v_readlane/v_writelane
SGPR spills, s_waitcnt vmcnt(0))."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("ds_read", "ds_load")): return "lds_rd"
    if op.startswith(("ds_write", "ds_store")): return "lds_wr"
    if op.startswith("buffer_load") or op.startswith("global_load"): return "vmem_ld"
    if op.startswith(("buffer_store", "global_store")): return "vmem_st"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    lines = open(path).read().split("\n")
    # find the kernel body
    start = next(i for i, l in enumerate(lines) if re.match(r"^(_Z\S*):", l) and pat in l)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
    body = []          # (kind, text): kind = 'label' | 'ins'
    for l in lines[start + 1:end]:
        s = l.strip()
        if not s or s.startswith((";", ".")) and not re.match(r"^\.LBB\S*:", s):
            continue
        if re.match(r"^\.LBB\S*:", s):
            body.append(("label", s.split(":")[0]))
        elif not s.startswith("."):
            body.append(("ins", s.split(";")[0].strip()))
    labels = {t: i for i, (k, t) in enumerate(body) if k == "label"}
    for i, (k, t) in enumerate(body):
        if k != "ins" or not t.startswith(("s_cbranch", "s_branch")):
            continue
        tgt = t.split()[-1]
        if tgt in labels and labels[tgt] < i:
            ins = [x for kk, x in body[labels[tgt]:i + 1] if kk == "ins"]
            n_mfma = sum(1 for x in ins if x.startswith("v_mfma"))
            if n_mfma == 0:
                continue
            mix = collections.Counter(classify(x.split()[0]) for x in ins)
            print(f"loop {tgt} .. {t.split()[0]}: {len(ins)} instructions, {n_mfma} MFMAs, {len(ins) / n_mfma:.2f} per MFMA")
            print("   ", ", ".join(f"{k} {v} ({v / n_mfma:.2f})" for k, v in mix.most_common()))
            bad = [x for x in ins if x.startswith(("scratch_", "v_accvgpr", "v_readlane", "v_writelane")) or
                   re.match(r"s_waitcnt vmcnt\(0\)", x)]
            print("    suspicious:", collections.Counter(x.split()[0] + (" vmcnt(0)" if "vmcnt(0)" in x else "") for x in bad))
            if dump:
                for x in ins:
                    print("      ", x)


main()
