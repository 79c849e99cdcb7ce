#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace + separate FETCH_SIZE / WRITE_SIZE --pmc passes)
for one kernel and grid size into profiles/<tag>.md and profiles/traffic_latest.json.

usage: summarize_prof.py <tag> <trace_dir> <fetch_dir> <write_dir> <kernel-substring> <grid_threads> <windows>

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are
in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced streaming reads,
so the read side is doubled; WRITE_SIZE is exact for 16-B-per-lane stores (narrower widths are
uncalibrated -- the write side here is a few KB and immaterial)."""
import csv
import glob
import json
import statistics
import sys
from pathlib import Path

tag, trace_dir, fetch_dir, write_dir, kname, grid, windows = sys.argv[1:8]
grid, windows = int(grid), int(windows)
REPO = Path(__file__).resolve().parents[1]


def rows(d, suffix):
    f = glob.glob(f"{d}/**/*_{suffix}.csv", recursive=True)
    out = list(csv.DictReader(open(f[0]))) if f else []
    for r in out:                      # the kernel trace spells the grid per dimension
        if "Grid_Size" not in r and "Grid_Size_X" in r:
            r["Grid_Size"] = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    return out


tr = [r for r in rows(trace_dir, "kernel_trace") if kname in r["Kernel_Name"] and int(r["Grid_Size"]) == grid]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr]
allk = {}
for r in rows(trace_dir, "kernel_trace"):
    key = (r["Kernel_Name"], int(r["Grid_Size"]))
    allk.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)


def counter(d, name):
    v = [float(r["Counter_Value"]) for r in rows(d, "counter_collection")
         if kname in r["Kernel_Name"] and int(r["Grid_Size"]) == grid and r["Counter_Name"] == name]
    return statistics.mean(v) if v else None


fetch_kib, write_kib = counter(fetch_dir, "FETCH_SIZE"), counter(write_dir, "WRITE_SIZE")
one = tr[0] if tr else {}
hbm = None
if fetch_kib is not None and write_kib is not None:
    hbm = fetch_kib * 1024 * 2 + write_kib * 1024
lines = [f"# rocprofv3 summary `{tag}`", "",
         f"kernel `{kname}`, grid {grid} threads ({grid // 256} workgroups x 256), {windows} windows per launch", "",
         "| quantity | value |", "|---|---|",
         f"| launches in trace | {len(dur)} |",
         f"| mean / min / max duration (us) | {statistics.mean(dur):.1f} / {min(dur):.1f} / {max(dur):.1f} |" if dur else "| duration | n/a |",
         f"| VGPR / AGPR / SGPR / LDS bytes | {one.get('VGPR_Count')} / {one.get('Accum_VGPR_Count')} / {one.get('SGPR_Count')} / {one.get('LDS_Block_Size')} |",
         f"| FETCH_SIZE (KiB, raw counter, mean per launch) | {fetch_kib} |",
         f"| WRITE_SIZE (KiB, raw counter, mean per launch) | {write_kib} |",
         f"| HBM bytes per launch = 2 x FETCH x 1024 + WRITE x 1024 (gfx950 correction) | {hbm} |",
         "", "## all kernels in the trace (name, grid threads, calls, mean us)", ""]
for (n, g), v in sorted(allk.items(), key=lambda kv: -sum(kv[1])):
    lines.append(f"- `{n}` grid {g}: {len(v)} calls, mean {statistics.mean(v):.1f} us, total {sum(v) / 1e3:.2f} ms")
(REPO / "profiles").mkdir(exist_ok=True)
(REPO / "profiles" / f"{tag}.md").write_text("\n".join(lines) + "\n")
if hbm is not None:
    (REPO / "profiles" / "traffic_latest.json").write_text(json.dumps(
        {"tag": tag, "kernel": kname, "windows": windows, "hbm_bytes_per_launch": hbm,
         "fetch_kib_raw": fetch_kib, "write_kib_raw": write_kib, "kernel_us_mean": statistics.mean(dur) if dur else None}))
print("\n".join(lines))
