"""Gaps between consecutive kernels of a rocprofv3 --kernel-trace database (start of a kernel minus end of the one in front, same queue order):
python tools/trace_gaps.py <trace_dir> [skip_first_kernels]"""
import sqlite3, glob, sys, collections, statistics
d = sys.argv[1]; skip = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for f in glob.glob(d + "/**/*.db", recursive=True):
    con = sqlite3.connect(f)
    rows = sorted(con.execute("select start, end, name from kernels").fetchall())[skip:]
    gaps = collections.defaultdict(list)
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        gaps[(n0.split("(")[0][-40:], n1.split("(")[0][-40:])].append((s1 - e0) / 1e3)
    busy = sum(e - s for s, e, _ in rows) / 1e3; span = (rows[-1][1] - rows[0][0]) / 1e3
    print(f"{len(rows)} kernels, busy {busy:.0f} us of {span:.0f} us ({busy / span * 100:.1f} %)")
    for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:8]:
        print(f"  {len(v):5d} x gap {statistics.median(v):6.2f} us (p90 {sorted(v)[int(len(v) * 0.9)]:6.2f})   {k[0]} -> {k[1]}")
