#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace + separate FETCH_SIZE / WRITE_SIZE --pmc passes, optionally an MFMA /
wave-state pass) for one kernel and grid size into profiles/<tag>.md, and record the HBM bytes per launch in
profiles/traffic_latest.json (one entry per kernel; bench.py reads it for `roofline.traffic`).

usage: summarize_prof.py <tag> <trace_dir> <fetch_dir> <write_dir> <kernel-substring> <grid_threads> <windows>
                         [--pmc-dir DIR] [--source csrc/file.hip] [--lds BYTES] [--flop-per-launch F] [--peak-tflops P]
                         [--note TEXT]

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced streaming reads, so the read side is doubled;
WRITE_SIZE is exact for 16-B-per-lane stores.

Registers come from the CODE OBJECT, not from the trace: rocprofv3's per-dispatch `VGPR_Count` is the architectural part
only, `Accum_VGPR_Count` reads 0 and `LDS_Block_Size` omits dynamic LDS on this ROCm (round 1's "236 / 0 / 0" row).  With
--source the kernel's translation unit is compiled to assembly here (hipcc -S, no GPU needed) and `.vgpr_count` (the
unified total), `.agpr_count`, `.sgpr_count` and the scratch size are read from its metadata; --lds states the dynamic
LDS bytes the launcher passes (smem_bytes of the instantiation)."""
import argparse
import csv
import glob
import json
import re
import statistics
import subprocess
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
ap = argparse.ArgumentParser()
for a in ("tag", "trace_dir", "fetch_dir", "write_dir", "kname"):
    ap.add_argument(a)
ap.add_argument("grid", type=int)
ap.add_argument("windows", type=int)
ap.add_argument("--pmc-dir", action="append", default=[])
ap.add_argument("--source")
ap.add_argument("--lds", type=int)
ap.add_argument("--flop-per-launch", type=float)
ap.add_argument("--peak-tflops", type=float)
ap.add_argument("--note", default="")
ap.add_argument("--wg-threads", type=int, default=256)
ap.add_argument("--skip-first", type=int, default=0, help="leave the first N matching launches out of the duration mean (clock ramp)")
ap.add_argument("--count", type=int, default=0, help="only this many launches behind --skip-first (later launches of the command run under other conditions)")
ap.add_argument("--min-us", type=float, default=0.0, help="only launches at least this long (the same instantiation also serves shorter windows)")
ap.add_argument("--model", default=None, help="pocket | watch | uarm | ff | imupose: part of the traffic entry's key (the same kernel serves several models)")
ap.add_argument("--T", type=int, default=None, help="window length of the profiled launches: part of the traffic entry's key")
args = ap.parse_args()


def rows(d, suffix):
    """rows of the rocprofv3 output under `d`: CSV files (--output-format csv) or the rocpd SQLite database (the default
    of ROCm 7.2: views `kernels` and `counters_collection`), in the CSV column names"""
    out = []
    for f in glob.glob(f"{d}/**/*_{suffix}.csv", recursive=True):
        out += list(csv.DictReader(open(f)))
    for r in out:                      # the kernel trace spells the grid per dimension
        if "Grid_Size" not in r and "Grid_Size_X" in r:
            r["Grid_Size"] = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    if out:
        return out
    import sqlite3
    for f in glob.glob(f"{d}/**/*.db", recursive=True):
        con = sqlite3.connect(f)
        if suffix == "kernel_trace":
            for name, start, end, gx, gy, gz in con.execute("select name, start, end, grid_x, grid_y, grid_z from kernels"):
                out.append({"Kernel_Name": name, "Start_Timestamp": start, "End_Timestamp": end, "Grid_Size": gx * gy * gz})
        else:
            for name, grid, cname, val in con.execute("select kernel_name, grid_size, counter_name, value from counters_collection"):
                out.append({"Kernel_Name": name, "Grid_Size": grid, "Counter_Name": cname, "Counter_Value": val})
    return out


def mine(r):
    return args.kname in r["Kernel_Name"] and int(r["Grid_Size"]) == args.grid


trace = rows(args.trace_dir, "kernel_trace")
tr = sorted((r for r in trace if mine(r)), key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr]
dur = [d for d in dur if d >= args.min_us]
steady = dur[args.skip_first:] if len(dur) > args.skip_first else dur
if args.count:
    steady = steady[:args.count]
allk = {}
for r in trace:
    allk.setdefault((r["Kernel_Name"], int(r["Grid_Size"])), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)


def upper_cluster(v):
    """the same instantiation also runs shorter windows in the same command (stream bank, T = 6): keep the launches of the
    benchmark shape = the values within 2x of the largest"""
    if not v:
        return v
    top = max(v)
    return [x for x in v if x >= 0.5 * top]


def counter(d, name):
    v = [float(r["Counter_Value"]) for r in rows(d, "counter_collection") if mine(r) and r["Counter_Name"] == name]
    v = upper_cluster(v)
    return statistics.mean(v) if v else None


def code_object_resources():
    if not args.source:
        return None
    src = (REPO / "arm-pose-estimation_amd" / args.source).resolve()
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "k.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", str(src),
                        "-o", str(out)], check=True, capture_output=True, cwd=src.parent)
        text = out.read_text()
    want = re.sub(r"[^A-Za-z0-9_]", "", args.kname.split("<")[0])
    nums = [int(x) for x in re.findall(r"-?\d+", args.kname.split("<", 1)[1])] if "<" in args.kname else []
    best = None
    for blk in text.split("  - .agpr_count:")[1:]:
        blk = ".agpr_count:" + blk
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name or want not in name.group(1):
            continue
        mangled = name.group(1)
        # template arguments appear as Li<N>E / Lb<0|1>E in the mangled name, in order
        targs = [int(x) for x in re.findall(r"L[ib](\d+)E", mangled)]
        if nums and targs[:len(nums)] != nums:
            continue
        g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
        best = {"symbol": mangled, "agpr": g("agpr_count"), "vgpr_total": g("vgpr_count"), "sgpr": g("sgpr_count"),
                "scratch": g("private_segment_fixed_size"), "static_lds": g("group_segment_fixed_size")}
        break
    return best


fetch_kib, write_kib = counter(args.fetch_dir, "FETCH_SIZE"), counter(args.write_dir, "WRITE_SIZE")
hbm = fetch_kib * 1024 * 2 + write_kib * 1024 if fetch_kib is not None and write_kib is not None else None
res = code_object_resources()
mean_us = statistics.mean(steady) if steady else None
lines = [f"# rocprofv3 summary `{args.tag}`", "",
         f"kernel `{args.kname}`, grid {args.grid} threads ({args.grid // args.wg_threads} workgroups x {args.wg_threads}), {args.windows} windows per launch", ""]
if args.note:
    lines += [args.note, ""]
try:
    _info = json.loads((REPO / "arm-pose-estimation_amd" / "lib" / "build_info.json").read_text())
    _obj = Path(args.source).name.replace(".hip", ".o") if args.source else None
    lines += [f"Library built from commit `{_info.get('commit')}`" + (f"; object `{_obj}` sha256 `{str(_info['objects'].get(_obj))[:16]}...`" if _obj else ""), ""]
except Exception:
    pass
lines += ["| quantity | value |", "|---|---|", f"| launches in trace | {len(dur)} |"]
if dur:
    lines.append(f"| mean / min / max duration (us), launches {args.skip_first + 1}..{args.skip_first + len(steady) if args.count else ''} | {mean_us:.1f} / {min(steady):.1f} / {max(steady):.1f} |")
    if args.skip_first:
        lines.append(f"| mean of the first {args.skip_first} launches (clock ramp, us) | {statistics.mean(dur[:args.skip_first]):.1f} |")
if res:
    lines.append(f"| registers (code object `{res['symbol'][:60]}...`) | {res['vgpr_total'] - res['agpr']} VGPR + {res['agpr']} AGPR "
                 f"= {res['vgpr_total']} of the 512 unified, {res['sgpr']} SGPR, scratch {res['scratch']} B |")
if args.lds is not None:
    lines.append(f"| LDS per workgroup (dynamic, launch argument) | {args.lds} B = {args.lds / 1024:.1f} KiB of 160 |")
if args.flop_per_launch and mean_us:
    tf = args.flop_per_launch / (mean_us * 1e-6) / 1e12
    lines.append(f"| algorithmic FLOP per launch / mean duration | {args.flop_per_launch:.6g} / {mean_us:.1f} us = {tf:.1f} TFLOP/s"
                 + (f" = {tf / args.peak_tflops * 100:.1f} % of {args.peak_tflops:g}" if args.peak_tflops else "") + " |")
lines += [f"| FETCH_SIZE (KiB, raw counter, mean per launch) | {fetch_kib} |",
          f"| WRITE_SIZE (KiB, raw counter, mean per launch) | {write_kib} |",
          f"| HBM bytes per launch = 2 x FETCH x 1024 + WRITE x 1024 (gfx950 correction) | {hbm} |"]
pmc = {}
for d in args.pmc_dir:
    for r in rows(d, "counter_collection"):
        if mine(r):
            pmc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
if pmc:
    lines += ["", "## counters (separate `rocprofv3 --kernel-trace --pmc` passes of the same command; mean per launch)", "",
              "| counter | mean per launch |", "|---|---|"]
    m = {k: statistics.mean(upper_cluster(v)) for k, v in pmc.items()}
    for k in sorted(m):
        lines.append(f"| `{k}` | {m[k]:,.0f} |")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; busy cycles over the 1024 SIMDs
        util = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
        lines.append(f"| **MfmaUtil** = MFMA busy / (active cycles per XCD x 1024 SIMDs) | **{util * 100:.1f} %** |")
    if "SQ_WAVE_CYCLES" in m:
        for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
            if k in m:
                lines.append(f"| {k} / SQ_WAVE_CYCLES | {m[k] / m['SQ_WAVE_CYCLES'] * 100:.1f} % |")
lines += ["", "## all kernels in the trace (name, grid threads, calls, mean us)", ""]
for (n, g), v in sorted(allk.items(), key=lambda kv: -sum(kv[1])):
    lines.append(f"- `{n if len(n) <= 160 else n[:157] + '...'}` grid {g}: {len(v)} calls, mean {statistics.mean(v):.1f} us, total {sum(v) / 1e3:.2f} ms")
(REPO / "profiles").mkdir(exist_ok=True)
(REPO / "profiles" / f"{args.tag}.md").write_text("\n".join(lines) + "\n")
if hbm is not None:
    tfile = REPO / "profiles" / "traffic_latest.json"
    try:
        cur = json.loads(tfile.read_text())
        ents = cur.get("kernels", [cur])
    except Exception:
        ents = []
    # one entry per (kernel instantiation, windows, model, T): the pocket bank must not quote the watch bank's bytes
    same = lambda e: (e.get("kernel") == args.kname and e.get("windows") == args.windows and e.get("model") == args.model and e.get("T") == args.T)
    ents = [e for e in ents if not same(e)]
    ent = {"tag": args.tag, "kernel": args.kname, "windows": args.windows, "model": args.model, "T": args.T, "hbm_bytes_per_launch": hbm,
           "fetch_kib_raw": fetch_kib, "write_kib_raw": write_kib, "kernel_us_mean": mean_us}
    # which build was measured: the commit the library was built from and the SHA-256 of the kernel's object file
    # (lib/build_info.json, written by csrc/Makefile at link time); bench.py compares the hash with the library it runs
    try:
        info = json.loads((REPO / "arm-pose-estimation_amd" / "lib" / "build_info.json").read_text())
        ent["commit"] = info.get("commit")
        if args.source:
            obj = Path(args.source).name.replace(".hip", ".o")
            ent["object"] = obj
            ent["object_sha256"] = info["objects"].get(obj)
    except Exception:
        pass
    ents.append(ent)
    tfile.write_text(json.dumps({"kernels": ents}, indent=1))
print("\n".join(lines[:40]))
