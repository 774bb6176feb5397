"""DEV TOOL: per kernel name, the shader clock and MFMA-pipe occupancy of its dispatches in a rocprofv3 --pmc run
(GRBM_GUI_ACTIVE [+ SQ_VALU_MFMA_BUSY_CYCLES]; the MI355X guide's derivations: clock = GRBM_GUI_ACTIVE / 8 XCDs / duration).

    python tools/r05/pmc_clock.py <rocprof output dir> [label]
"""
import glob
import sqlite3
import sys
from collections import defaultdict


def main():
    out = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else out
    db = glob.glob(out + "/**/*_results.db", recursive=True)[0]
    c = sqlite3.connect(db)
    ev = c.execute("select name, dispatch_id, counter_name, sum(counter_value), max(duration) from pmc_events "
                   "group by name, dispatch_id, counter_name").fetchall()
    per = defaultdict(lambda: defaultdict(dict))
    for n, did, cn, val, d in ev:
        per[n][did][cn] = val
        per[n][did]["dur"] = d
    print(f"{label}:")
    rows = []
    for n, disp in per.items():
        if "flash_" not in n:
            continue
        ds = [v for v in disp.values() if "GRBM_GUI_ACTIVE" in v and v["dur"] > 0]
        if not ds:
            continue
        dur = sum(v["dur"] for v in ds) / len(ds) / 1e3
        clk = sum(v["GRBM_GUI_ACTIVE"] / 8 / v["dur"] for v in ds) / len(ds)
        busy = None
        if all("SQ_VALU_MFMA_BUSY_CYCLES" in v for v in ds):
            busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * (v["GRBM_GUI_ACTIVE"] / 8)) for v in ds) / len(ds)
        rows.append((dur * len(ds), n.replace("void ", "").split("(")[0], len(ds), dur, clk, busy))
    for _, n, k, dur, clk, busy in sorted(rows, reverse=True):
        print(f"   {n[:56]:56s} dispatches {k:4d}  mean {dur:8.1f} us  clock {clk:5.2f} GHz" +
              (f"  MFMA pipe busy {busy * 100:5.1f} %  -> cycles per dispatch {dur * clk * 1e3:10.0f}" if busy is not None else ""))


if __name__ == "__main__":
    main()
