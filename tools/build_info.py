#!/usr/bin/env python3
"""Written by csrc/Makefile behind every link of libape_hip.so: lib/build_info.json = the commit the tree was built from
(`git describe --always --dirty`; "unknown" where there is no git, e.g. a rebuild on the GPU box) and the SHA-256 of every object
file.  tools/summarize_prof.py stamps each profiles/traffic_latest.json entry with the commit and with the hash of the kernel's
object; bench.py compares that hash with the library it runs and prints `traffic_stale: true` when they differ."""
import hashlib, json, subprocess, sys
from pathlib import Path
lib = Path(sys.argv[1])
try:
    commit = subprocess.run(["git", "describe", "--always", "--dirty"], capture_output=True, text=True, cwd=lib, check=True).stdout.strip()
except Exception:
    commit = "unknown"
objs = {Path(o).name: hashlib.sha256(Path(o).read_bytes()).hexdigest() for o in sys.argv[2:]}
old = {}
try:
    old = json.loads((lib / "build_info.json").read_text())
except Exception:
    pass
# an unchanged set of objects keeps the commit it was first built from -- a re-link on a box without git must not say
# "unknown" -- unless the tree has been committed since: a clean describe replaces a "-dirty" one
if old.get("objects") == objs:
    prev = old.get("commit", "unknown")
    if commit == "unknown" or (prev != "unknown" and not prev.endswith("-dirty")) or commit.endswith("-dirty"):
        commit = prev if prev != "unknown" else commit
(lib / "build_info.json").write_text(json.dumps({"commit": commit, "objects": objs}, indent=1) + "\n")
