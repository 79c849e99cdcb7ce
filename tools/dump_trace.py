"""durations of the kernels whose name contains argv[2], in launch order, from a rocprofv3 --kernel-trace output directory argv[1]"""
import glob, sqlite3, sys
for f in glob.glob(f"{sys.argv[1]}/**/*.db", recursive=True):
    con = sqlite3.connect(f)
    rows = sorted(con.execute("select start, end, name, grid_x from kernels"))
    sel = [(i, r) for i, r in enumerate(rows) if sys.argv[2] in r[2]]
    print(len(rows), "kernels,", len(sel), "matching")
    for i, r in sel[:int(sys.argv[3]) if len(sys.argv) > 3 else 12]:
        prev = rows[i - 1] if i else None
        print(f"#{i}: {(r[1] - r[0]) / 1e3:10.1f} us grid {r[3]}  gap to previous {(r[0] - prev[1]) / 1e3 if prev else 0:8.1f} us  previous: {prev[2][:60] if prev else ''}")
