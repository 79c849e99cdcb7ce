set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_bank
rocprofv3 --kernel-trace --stats -d /tmp/prof_bank -- python3 $R/tests/tools/bank_trace.py $1 $2 $3 > $R/gpurun_out/bank_trace.log 2>&1
cd $R
python3 - <<'PY'
import sqlite3, glob, collections
for f in glob.glob("/tmp/prof_bank/**/*.db", recursive=True):
    con = sqlite3.connect(f)
    d = collections.defaultdict(list)
    for name, start, end, gx in con.execute("select name, start, end, grid_x from kernels"):
        d[(name[:70], gx)].append((end - start) / 1e3)
    for (name, gx), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        if len(v) >= 20: print(f"{len(v):5d} x {sum(v)/len(v):8.1f} us  grid {gx:8d}  {name}")
PY
tail -1 $R/gpurun_out/bank_trace.log
