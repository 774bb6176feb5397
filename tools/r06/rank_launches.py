"""DEV TOOL: the flash launches of ONE iteration of an emulated rank in issue order (rocprofv3 --kernel-trace of
tools/rank_emulation.py): name, work items (grid), start offset and duration -- what the link model's schedule replay is fed with.

    python tools/r06/rank_launches.py <rocprof output dir> <iterations in the trace>
"""
import glob
import sqlite3
import sys


def main():
    out, iters = sys.argv[1], int(sys.argv[2])
    db = glob.glob(out + "/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    rows = c.execute("select S.display_name, K.start, K.end, K.grid_size_x, K.workgroup_size_x from rocpd_kernel_dispatch K "
                     "join rocpd_info_kernel_symbol S on S.id = K.kernel_id and S.guid = K.guid order by K.start").fetchall()
    flash = [r for r in rows if "flash_" in r[0] or "split_merge" in r[0] or "reduce_" in r[0] or "delta_kernel" in r[0]]
    per = len(flash) // iters
    last = flash[-per:]
    t0 = last[0][1]
    print(f"{len(flash)} flash-path launches in {iters} iterations = {per} per iteration; the last iteration:")
    for name, s, e, gx, wx in last:
        short = name.replace("void ", "").replace("usp::", "").split("(")[0][:44]
        print(f"   +{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  wgs {gx // max(1, wx):5d}  {short}")
    print(f"   window {(last[-1][2] - t0) / 1e6:.3f} ms, kernel time {sum(e - s for _, s, e, _, _ in last) / 1e6:.3f} ms")


if __name__ == "__main__":
    main()
