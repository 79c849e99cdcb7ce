set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "41 25 pocket" "21 50 uarm"; do
  set -- $spec
  rm -rf /tmp/tr_small
  rocprofv3 --kernel-trace --stats -d /tmp/tr_small -- python3 $R/tests/tools/bank_trace.py $1 $2 300 auto check $3 > $R/gpurun_out/small_$3.out 2>/dev/null
  grep -v amdgpu $R/gpurun_out/small_$3.out
  python3 - /tmp/tr_small <<'PY'
import sqlite3, glob, collections, sys
for f in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    con = sqlite3.connect(f)
    d = collections.defaultdict(list)
    for name, start, end, gx in con.execute("select name, start, end, grid_x from kernels"):
        d[(name[:100], gx)].append((end - start) / 1e3)
    for (name, gx), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        if len(v) >= 100: print(f"{len(v):6d} x {sum(v[len(v)//2:])/len(v[len(v)//2:]):9.1f} us (min {min(v):8.1f})  grid {gx:8d}  {name}")
PY
done
