"""DEV TOOL: what runs BESIDE the flash kernels in a rocprofv3 --kernel-trace run, and what the timeline's gaps are.

    python tools/r05/trace_overlap.py <rocprof output dir> [label] [skip_first_fraction]

Per flash kernel name: calls, mean duration, and the mean fraction of a dispatch's duration during which (a) a copy kernel
(__amd_rocclr_copyBuffer / copy_rows / fill), (b) another flash kernel was in flight on another stream; then the stretch of
the timeline that is covered by no flash kernel at all (launch gaps + waits).  The first `skip_first_fraction` of the trace
(warm-up iterations) is ignored.
"""
import glob
import sqlite3
import sys
from collections import defaultdict


def main():
    out = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else out
    skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.4
    db = glob.glob(out + "/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    rows = c.execute("select S.display_name, K.start, K.end, K.stream_id from rocpd_kernel_dispatch K "
                     "join rocpd_info_kernel_symbol S on S.id = K.kernel_id and S.guid = K.guid order by K.start").fetchall()
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    cut = t0 + skip * (t1 - t0)
    rows = [r for r in rows if r[1] >= cut]
    flash = [r for r in rows if "flash_" in r[0]]
    copies = [r for r in rows if "copyBuffer" in r[0] or "copy_rows" in r[0] or "fill" in r[0].lower()]

    def overlap(a, others):
        tot = 0
        for o in others:
            if o[2] <= a[1] or o[1] >= a[2] or o is a:
                continue
            tot += min(a[2], o[2]) - max(a[1], o[1])
        return tot
    agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for k in flash:
        d = k[2] - k[1]
        name = k[0].replace("void ", "").split("(")[0]
        a = agg[name]
        a[0] += 1
        a[1] += d
        a[2] += overlap(k, copies) / d
        a[3] += overlap(k, flash) / d
    print(f"{label}: {len(rows)} dispatches behind the warm-up cut, {len(flash)} flash launches, {len(copies)} copies")
    for name, (n, d, oc, of) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"   {name[:56]:56s} calls {n:4d}  mean {d / n / 1e3:8.1f} us   beside copies {oc / n * 100:5.1f} %   beside another flash kernel {of / n * 100:5.1f} %")
    # union of the flash intervals against the whole window
    iv = sorted((k[1], k[2]) for k in flash)
    covered, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            covered += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    covered += cur_e - cur_s
    span = iv[-1][1] - iv[0][0]
    print(f"   flash kernels cover {covered / 1e6:.2f} of {span / 1e6:.2f} ms of the window ({(1 - covered / span) * 100:.1f} % uncovered)")


if __name__ == "__main__":
    main()
