#!/usr/bin/env python3
"""hipcc cannot see inside inline asm, so it inserts no wait states between a VALU instruction it emits and an asm MFMA that reads the
VALU's result as SrcA / SrcB (gfx950 needs them; a stale operand goes unnoticed by everything but a parity test).  This scans the
device assembly of a kernel file for that pattern: a VALU write (v_mov, v_accvgpr_read, any v_* but MFMA) of a register that one of the
next `window` instructions, an MFMA, reads as its first or second source.
    tools/check_mfma_hazards.py file.hip [extra hipcc flags]     exit code 1 when a hazard is found"""
import re, subprocess, sys, tempfile

def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()

def scan(path, window=2):
    ins = []
    for n, line in enumerate(open(path), 1):
        t = line.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"): continue
        ins.append((n, t.split(";")[0].strip()))
    bad = []
    for i, (n, t) in enumerate(ins):
        if not t.startswith("v_mfma") and not t.startswith("v_smfmac"): continue
        ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
        src = regs(ops[1]) | regs(ops[2])
        for k in range(1, window + 1):
            if i - k < 0: break
            pn, pt = ins[i - k]
            if pt.startswith(("v_mfma", "v_smfmac")) or not pt.startswith("v_"): continue
            dst = regs(pt.split(None, 1)[1].split(",")[0].strip())
            if dst & src: bad.append((pn, pt, n, t))
    return bad

def scan_early_reads(path, window=12):
    """a non-MFMA instruction that READS the destination of an asm MFMA within `window` instructions behind it (following branches), with
    no drain (s_nop 15) in between: the compiler, which takes the asm's result for ready, put a copy there (phi moves at a branch merge do it)"""
    ins, labels = [], {}
    for n, line in enumerate(open(path), 1):
        t = line.strip()
        if t.endswith(":") and not t.startswith(";"):
            labels[t[:-1]] = len(ins)
            continue
        if not t or t.startswith((";", ".", "//")): continue
        ins.append((n, t.split(";")[0].strip()))
    bad = set()
    def walk(i, left, dst, origin):
        while left > 0 and i < len(ins):
            qn, qt = ins[i]
            if qt.startswith("s_nop 15") or qt.startswith(("s_endpgm", "s_setpc")): return
            if qt.startswith("s_branch"):
                i = labels.get(qt.split()[1], len(ins)); continue
            if qt.startswith("s_cbranch"):
                walk(labels.get(qt.split()[1], len(ins)), left - 1, dst, origin)
            elif qt.startswith("v_mfma"):
                if regs(qt.split(None, 1)[1].split(",")[0].strip()) == dst: return       # the chain goes on: checked from there
            elif not qt.startswith("s_") and " " in qt:
                ops = [o.strip().split(" ")[0] for o in qt.split(None, 1)[1].split(",")]
                srcs = set()
                for o in (ops if qt.startswith(("ds_write", "buffer_store", "global_store")) else ops[1:]): srcs |= regs(o)
                if srcs & dst: bad.add(origin + (qn, qt))
            i += 1; left -= 1
    for i, (n, t) in enumerate(ins):
        if t.startswith("v_mfma"): walk(i + 1, window, regs(t.split(None, 1)[1].split(",")[0].strip()), (n, t))
    return sorted(bad)

if __name__ == "__main__":
    src = sys.argv[1]
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", *sys.argv[2:], src, "-o", f.name],
                       check=True, stderr=subprocess.DEVNULL)
        bad = scan(f.name)
        early = scan_early_reads(f.name) if "asm" in open(src).read() else []
    for pn, pt, n, t in bad: print(f"{src}: line {pn}: {pt}   ->   line {n}: {t}")
    for n, t, qn, qt in early: print(f"{src}: line {n}: {t}   read early by   line {qn}: {qt}")
    print(f"{src}: {len(bad)} VALU-write -> MFMA SrcA/SrcB adjacencies, {len(early)} early reads of an MFMA result")
    sys.exit(1 if bad or early else 0)
