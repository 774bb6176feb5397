"""DEV TOOL: the top kernels of a rocprofv3 --kernel-trace --stats run, per iteration.

    python tools/r05/trace_top.py <rocprof output dir> <iterations traced> [label] [rows]
"""
import glob
import sqlite3
import sys


def main():
    out, iters = sys.argv[1], float(sys.argv[2])
    label = sys.argv[3] if len(sys.argv) > 3 else out
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 12
    db = glob.glob(out + "/**/*_results.db", recursive=True)[0]
    rows = sqlite3.connect(db).execute("select name,total_calls,total_duration,average from top_kernels").fetchall()
    tot = sum(r[2] for r in rows)
    print("%s: total kernel time %.2f ms per iteration (%g iterations traced)" % (label, tot / iters / 1e3, iters))
    for name, calls, t, avg in rows[:n]:
        print("   %-70s calls %5d  total %9.1f us  avg %8.1f  (%.2f ms / iteration)"
              % (name.replace("void ", "")[:70], calls, t, avg, t / iters / 1e3))


if __name__ == "__main__":
    main()
