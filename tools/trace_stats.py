"""per-kernel median / mean duration and the median gap in front of each kernel, from a rocprofv3 --kernel-trace output directory:
python tools/trace_stats.py <dir> [skip-first-N launches per kernel]"""
import glob, re, sqlite3, statistics, sys
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for f in glob.glob(f"{sys.argv[1]}/**/*.db", recursive=True):
    con = sqlite3.connect(f)
    rows = sorted(con.execute("select start, end, name, grid_x from kernels"))
    by = {}
    for i, r in enumerate(rows):
        mm = re.search(r"(ape_\w+(<[^>]*>)?)", r[2])
        name = mm.group(1) if mm else r[2][:70]
        by.setdefault(name, []).append(((r[1] - r[0]) / 1e3, (r[0] - rows[i - 1][1]) / 1e3 if i else 0.0))
    print(f"{len(rows)} launches")
    for name, v in sorted(by.items(), key=lambda kv: -sum(d for d, _ in kv[1])):
        v = v[skip:] if len(v) > skip + 4 else v
        d = [x[0] for x in v]; g = [x[1] for x in v]
        print(f"  {name:72s} n={len(v):5d}  median {statistics.median(d):8.2f} us  mean {statistics.mean(d):8.2f}  median gap in front {statistics.median(g):7.2f} us")
