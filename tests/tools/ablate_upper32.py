"""What each part of a section of the upper-layer kernel of the Monte-Carlo bank (lstm_upper32.hip) costs: the bank frame with the
part switched off (ablation library only -- `make -C arm-pose-estimation_amd/csrc ablate`; outputs of the ablated runs are wrong by
construction).  python tests/tools/ablate_upper32.py [S] [n_mc]  (spawns one process per switch: the library reads APE_UP32_ABLATE)"""
import os, subprocess, sys
S = sys.argv[1] if len(sys.argv) > 1 else "1024"
n_mc = sys.argv[2] if len(sys.argv) > 2 else "32"
NOEX, NOACT, NOMFMA, NOX, NOBAR = 0x40000000, 0x20000000, 0x04000000, 0x00200000, 0x00100000
lib = "/root/repo/arm-pose-estimation_amd/lib/diag/libape_hip_ablate.so"
for tag, fl in (("everything on", 0), ("no h exchange (publish, flags, gather)", NOEX), ("no gate transcendentals", NOACT),
                ("no x copies", NOX), ("no exchange, no x copies", NOEX | NOX), ("no exchange / x / section barrier + wait", NOEX | NOX | NOBAR),
                ("no exchange / x / barrier / gates", NOEX | NOX | NOBAR | NOACT), ("no MFMAs", NOMFMA),
                ("skeleton (none of them)", NOEX | NOX | NOBAR | NOACT | NOMFMA)):
    env = dict(os.environ, APE_HIP_LIB=lib, APE_UP32_ABLATE=hex(fl))
    out = subprocess.run([sys.executable, "/root/repo/tests/tools/bank_trace.py", S, n_mc, "30", "auto", "nocheck"], env=env, capture_output=True, text=True)
    print(f"{tag:48s} {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]}", flush=True)
