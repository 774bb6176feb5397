#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases (gpurun_out/prof_*/**.db) into small text files for profiles/.

usage: tools/prof_summary.py <prof_dir> <out.txt>
  * every *_results.db: per-kernel call count / total / average duration (the --stats view);
  * databases with PMC events: per kernel, counter values summed over instances per dispatch and
    averaged over dispatches; HBM bytes derived as the MI355X guide prescribes
    (FETCH_SIZE, WRITE_SIZE are in KiB; FETCH_SIZE x2 on gfx950 for wide coalesced reads).
"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    return name if len(name) < 90 else name[:87] + "..."


def main():
    prof_dir, out_path = sys.argv[1], sys.argv[2]
    lines = []
    for db in sorted(glob.glob(os.path.join(prof_dir, "**", "*_results.db"), recursive=True)):
        rel = os.path.relpath(db, prof_dir)
        c = sqlite3.connect(db)
        lines.append(f"==== {rel}")
        try:
            rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
        except sqlite3.Error as e:
            lines.append(f"  (no top_kernels view: {e})")
            rows = []
        lines.append(f"  {'kernel':90s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'%':>6s}")
        for n, calls, tot, avg, pct in rows[:8]:
            lines.append(f"  {short(n):90s} {calls:6d} {tot:12.1f} {avg:10.2f} {pct:6.2f}")
        try:
            ev = c.execute("select name, dispatch_id, counter_name, sum(counter_value), max(duration) "
                           "from pmc_events group by name, dispatch_id, counter_name").fetchall()
        except sqlite3.Error:
            ev = []
        if ev:
            agg = defaultdict(lambda: defaultdict(list))
            dur = defaultdict(list)
            for n, did, cn, val, d in ev:
                agg[n][cn].append(val)
                dur[n].append(d)
            for n in agg:
                if "rocclr" in n:
                    continue
                lines.append(f"  PMC {short(n)}  (dispatches {len(next(iter(agg[n].values())))}, avg dur "
                             f"{sum(dur[n]) / len(dur[n]) / 1e3:.1f} us)")
                cv = {cn: sum(v) / len(v) for cn, v in agg[n].items()}
                for cn in sorted(cv):
                    lines.append(f"      {cn:28s} {cv[cn]:18.1f}")
                if "SQ_VALU_MFMA_BUSY_CYCLES" in cv and "SQ_BUSY_CYCLES" in cv and cv["SQ_BUSY_CYCLES"]:
                    lines.append(f"      -> MFMA busy / SQ busy cycles     = {cv['SQ_VALU_MFMA_BUSY_CYCLES'] / cv['SQ_BUSY_CYCLES']:.3f}"
                                 f"   (per-SE/XCD instance sums; ratio is what matters)")
                if "SQ_LDS_BANK_CONFLICT" in cv and cv.get("SQ_LDS_IDX_ACTIVE"):
                    lines.append(f"      -> LDS bank-conflict / LDS active = {cv['SQ_LDS_BANK_CONFLICT'] / cv['SQ_LDS_IDX_ACTIVE']:.4f}")
                if "SQ_WAVE_CYCLES" in cv and cv["SQ_WAVE_CYCLES"]:
                    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                        if k in cv:
                            lines.append(f"      -> {k} / SQ_WAVE_CYCLES = {cv[k] / cv['SQ_WAVE_CYCLES']:.3f}")
                if "FETCH_SIZE" in cv:
                    lines.append(f"      -> HBM read bytes/launch  = {cv['FETCH_SIZE'] * 1024 * 2 / 1e6:.1f} MB "
                                 f"(FETCH_SIZE KiB x2 gfx950 correction; raw {cv['FETCH_SIZE'] * 1024 / 1e6:.1f} MB)")
                if "WRITE_SIZE" in cv:
                    lines.append(f"      -> HBM write bytes/launch = {cv['WRITE_SIZE'] * 1024 / 1e6:.1f} MB (WRITE_SIZE KiB, uncalibrated)")
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
