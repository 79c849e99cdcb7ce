#!/usr/bin/env python3
"""hipcc cannot see inside inline asm, so it inserts no wait states between a VALU instruction it emits and an asm MFMA that reads the
VALU's result as SrcA / SrcB / SrcC (gfx950 needs them; a stale operand goes unnoticed by everything but a parity test).  This scans the
device assembly of a kernel file for that pattern: a VALU write (v_mov, v_accvgpr_read, any v_* but MFMA) of a register that one of the
next `window` instructions, an MFMA, reads as a source.
A third pattern, found in round 4 (lstm_mc_small.hip's diagnostic build: every launch aborted): a VALU instruction that writes a SCALAR
register (v_readlane_b32 reloading a spilled buffer descriptor, v_readfirstlane_b32, a v_cmp with an SGPR destination) needs 5 wait states
before a vector-memory instruction reads that register; hipcc pads its own loads and stores, not the ones inside an asm statement.
A fourth pattern, found in round 5 (lstm_upper128.hip: whole 32-row tiles off by 1e-4 .. 1e-3 on a cold start or beside a memory-bound
kernel): a vector-memory LOAD issued inside an asm statement with a compiler-allocated destination ("=v") so that it can stay in flight
under the MFMA stream, its value first touched by a later asm `s_waitcnt vmcnt(0)`.  hipcc takes the asm's output for valid at once and is
free to COPY the destination registers (a phi move at a branch merge did) or, where it can prove the value dead, to RE-USE them, in front
of that wait: the copy then holds what the registers held before, and a late-landing load clobbers the re-user's data.  The scan walks
the code from every such load (following branches, up to the first `s_waitcnt vmcnt(0)` on each path) and reports any instruction that
reads or writes a destination register.
A fifth pattern, found in round 5 (lstm_upper32.hip, layer-0 form: beside a memory-bound neighbour one frame in ~2000 of a fresh bank had
the integer 2 or 4 -- the slice epoch -- in word 0 of sixteen lanes of one wave's published slice): a vector-memory STORE of more than 64
bits issued inside an asm statement, its data registers written by a VALU instruction fewer than 2 wait states later (there: the epoch's
`v_add_u32 v2, 1, v51` right behind an untaken branch, v[2:5] being the store's data).  The store reads its data late (gfx940 and later: 2
wait states; none if a buffer store's soffset is an SGPR -- LLVM's GCNHazardRecognizer, VMEM store-data hazard); hipcc pads the stores it
emits itself, an asm statement is opaque to it.
A sixth pattern, named by the round-5 code review before it was ever seen (round 6): the LDS-DMA statements write M0 inside the asm
(`s_mov_b32 m0, %0`) and hipcc does not know -- listing "m0" as a clobber changes no code and draws a "reserved register" warning.  A
compiler-emitted instruction that READS M0 (s_movrel / v_movrel indexing, its own `buffer_load ... lds`, s_sendmsg, ds_gws, an explicit m0
operand) while the last writer of M0 in layout order was an asm statement would use the asm's LDS address as its index.  The kernels give the
compiler no reason to use M0 (no dynamically indexed register arrays in the loops that hold the DMA statements); the scan makes that a
checked fact of every build.
A seventh, from the same table: gfx940 and later need one wait state between a transcendental VALU instruction
(v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos) and a non-transcendental VALU instruction that reads its result; hipcc pads its own
consumers, not a VALU instruction inside an asm statement (the packed multiply / max of mlp_pipe.hip, the permlane swaps, an MFMA).
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
        src_c = regs(ops[3]) if len(ops) > 3 else set()        # the accumulator's start value: flagged when written by the instruction right in front
        for k in range(1, window + 1):
            if i - k < 0: break
            pn, pt = ins[i - k]
            if pt.startswith(("v_mfma", "v_smfmac")) or not pt.startswith("v_"): continue
            dst = regs(pt.split(None, 1)[1].split(",")[0].strip())
            if dst & (src | src_c if k == 1 else src): bad.append((pn, pt, n, t))
    return bad

def sregs(tok):
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    return {int(m.group(1))} if m else set()

def scan_sgpr_into_asm_vmem(path, need=5):
    """a vector-memory instruction INSIDE an asm statement (between ;;#ASMSTART and ;;#ASMEND) that reads a scalar register a VALU
    instruction wrote fewer than `need` wait states earlier (an s_nop k counts k + 1, any other instruction 1; a label or branch ends
    the look-back: straight-line code only, which is where a spill reload sits)"""
    ins, in_asm = [], False
    for n, line in enumerate(open(path), 1):
        t = line.strip()
        if t.startswith(";;#ASMSTART"): in_asm = True; continue
        if t.startswith(";;#ASMEND"): in_asm = False; continue
        if t.split(";")[0].strip().endswith(":") and not t.startswith(";"):
            ins.append((n, "LABEL", False)); continue
        if not t or t.startswith((";", ".", "//")): continue
        ins.append((n, t.split(";")[0].strip(), in_asm))
    bad = []
    for i, (n, t, a) in enumerate(ins):
        if not a or not t.startswith(("buffer_", "global_", "flat_", "scratch_")) or " " not in t: continue
        used = set()
        for o in t.split(None, 1)[1].replace(",", " ").split(): used |= sregs(o)
        if not used: continue
        states, k = 0, i - 1
        while k >= 0 and states < need:
            pn, pt, _ = ins[k]
            if pt == "LABEL" or pt.startswith(("s_branch", "s_cbranch", "s_barrier")): break
            if pt.startswith("v_") and " " in pt:
                dst = sregs(pt.split(None, 1)[1].split(",")[0].strip())
                if dst & used: bad.append((pn, pt, n, t, states))
            states += (int(pt.split()[1]) + 1) if pt.startswith("s_nop") else 1
            k -= 1
    return bad

def scan_early_reads(path, window=12):
    """a non-MFMA instruction that READS the destination of an asm MFMA within `window` instructions behind it (following branches), with
    no drain (s_nop 15) in between: the compiler, which takes the asm's result for ready, put a copy there (phi moves at a branch merge do it)"""
    ins, labels = [], {}
    for n, line in enumerate(open(path), 1):
        t = line.strip()
        c = t.split(";")[0].strip()                    # (a label inside a loop carries a trailing comment: `.LBB0_104:   ; in Loop: ...`)
        if c.endswith(":") and not t.startswith(";"):
            labels[c[:-1]] = len(ins)
            continue
        if not t or t.startswith((";", ".", "//")): continue
        ins.append((n, c))
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

def vregs_all(text):
    out = set()
    for a, b, c in re.findall(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        out |= {int(c)} if c else set(range(int(a), int(b) + 1))
    return out

def scan_async_asm_loads(path, limit=6000):
    """asm-issued VMEM loads with a VGPR destination (not LDS-DMA): every instruction between the load and the first `s_waitcnt vmcnt(0)` on
    each path that touches a destination register"""
    ins, labels, in_asm = [], {}, False
    for n, line in enumerate(open(path), 1):
        t = line.strip()
        if t.startswith(";;#ASMSTART"): in_asm = True; continue
        if t.startswith(";;#ASMEND"): in_asm = False; continue
        c = t.split(";")[0].strip()
        if c.endswith(":") and not t.startswith(";"):
            labels[c[:-1]] = len(ins); continue
        if not t or t.startswith((";", ".", "//")): continue
        ins.append((n, c, in_asm))
    bad = set()
    for i, (n, t, a) in enumerate(ins):
        if not a or not t.startswith(("global_load", "buffer_load", "flat_load")) or t.rstrip().endswith(" lds") or " lds " in t: continue
        dst = regs(t.split(None, 1)[1].split(",")[0].strip())
        if not dst: continue
        seen, stack = set(), [(i + 1, limit)]
        while stack:
            j, left = stack.pop()
            while left > 0 and j < len(ins) and j not in seen:
                seen.add(j)
                qn, qt, qa = ins[j]
                if qt.startswith("s_waitcnt") and "vmcnt(0)" in qt: break
                if qt.startswith(("s_endpgm", "s_setpc")): break
                if qt.startswith("s_branch"):
                    j = labels.get(qt.split()[1], len(ins)); continue
                if qt.startswith("s_cbranch"):
                    stack.append((labels.get(qt.split()[1], len(ins)), left - 1))
                elif " " in qt and not qt.startswith("s_") and vregs_all(qt.split(None, 1)[1]) & dst:
                    bad.add((n, t, qn, qt))
                j += 1; left -= 1
    return sorted(bad)

def scan_asm_wide_stores(path, need=2):
    """asm-issued stores of more than 64 bits (a buffer store only with a literal soffset): a VALU write of a data register within `need`
    wait states behind it, on any path (an s_nop k counts k + 1, any other instruction 1)"""
    ins, labels, in_asm = [], {}, False
    for n, line in enumerate(open(path), 1):
        t = line.strip()
        if t.startswith(";;#ASMSTART"): in_asm = True; continue
        if t.startswith(";;#ASMEND"): in_asm = False; continue
        c = t.split(";")[0].strip()
        if c.endswith(":") and not t.startswith(";"):
            labels[c[:-1]] = len(ins); continue
        if not t or t.startswith((";", ".", "//")): continue
        ins.append((n, c, in_asm))
    bad = set()
    for i, (n, t, a) in enumerate(ins):
        m = re.match(r"(buffer|global|flat|scratch)_store_dwordx[34]\b", t)
        if not a or not m: continue
        ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
        if m.group(1) == "buffer" and len(ops) > 3 and re.match(r"s\d+|s\[|m0|ttmp", ops[3].split()[0]): continue      # soffset in a register: no hazard
        data = regs(ops[0])
        stack, seen = [(i + 1, 0)], set()
        while stack:
            j, states = stack.pop()
            while states < need and j < len(ins) and (j, states) not in seen:
                seen.add((j, states))
                qn, qt, _ = ins[j]
                if qt.startswith(("s_endpgm", "s_setpc")): break
                if qt.startswith("s_branch"):
                    j = labels.get(qt.split()[1], len(ins)); states += 1; continue
                if qt.startswith("s_cbranch"):
                    stack.append((labels.get(qt.split()[1], len(ins)), states + 1))
                elif qt.startswith("v_") and " " in qt and regs(qt.split(None, 1)[1].split(",")[0].strip()) & data:
                    bad.add((n, t, qn, qt, states))
                states += (int(qt.split()[1]) + 1) if qt.startswith("s_nop") else 1
                j += 1
    return sorted(bad)

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")

def scan_trans_into_asm_valu(path):
    """a VALU instruction INSIDE an asm statement that reads, as the very next instruction, the result of a transcendental one"""
    ins, in_asm = [], False
    for n, line in enumerate(open(path), 1):
        t = line.strip()
        if t.startswith(";;#ASMSTART"): in_asm = True; continue
        if t.startswith(";;#ASMEND"): in_asm = False; continue
        c = t.split(";")[0].strip()
        if not c or c.startswith((".", "//")) or c.endswith(":"): continue
        ins.append((n, c, in_asm))
    bad = []
    for i in range(1, len(ins)):
        n, t, a = ins[i]
        pn, pt, _ = ins[i - 1]
        if not a or not t.startswith("v_") or " " not in t or not pt.startswith(TRANS) or " " not in pt: continue
        dst = regs(pt.split(None, 1)[1].split(",")[0].strip())
        srcs = set()
        for o in t.split(None, 1)[1].split(",")[1:]: srcs |= regs(o.strip().split(" ")[0])
        if t.startswith(("v_mfma", "v_smfmac")): srcs |= regs(t.split(None, 1)[1].split(",")[0].strip())
        if dst & srcs: bad.append((pn, pt, n, t))
    return bad

M0_IMPLICIT = ("s_movrel", "v_movrel", "s_sendmsg", "ds_gws", "v_interp", "ds_append", "ds_consume", "ds_ordered_count")

def scan_m0_after_asm(path):
    """a compiler-emitted instruction that reads M0 -- explicitly, or implicitly (M0_IMPLICIT, its own LDS-DMA loads) -- while the last write
    of M0 in layout order came from inside an asm statement (labels and branches do not reset the state: conservative)"""
    ins, in_asm = [], False
    for n, line in enumerate(open(path), 1):
        t = line.strip()
        if t.startswith(";;#ASMSTART"): in_asm = True; continue
        if t.startswith(";;#ASMEND"): in_asm = False; continue
        c = t.split(";")[0].strip()
        if not c or c.startswith((".", "//")) or c.endswith(":"): continue
        ins.append((n, c, in_asm))
    bad, writer = [], None
    for n, t, a in ins:
        parts = t.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        toks = re.findall(r"[A-Za-z_][A-Za-z_0-9]*", parts[1]) if len(parts) > 1 else []
        writes = bool(ops) and ops[0] == "m0" and parts[0].startswith("s_")
        reads = ("m0" in toks[1:] if writes else "m0" in toks) or parts[0].startswith(M0_IMPLICIT) or \
                (parts[0].startswith(("buffer_load", "global_load")) and (t.endswith(" lds") or " lds " in t or parts[0].startswith("global_load_lds")))
        if not a and reads and writer is not None and writer[0]: bad.append((writer[1], writer[2], n, t))
        if writes: writer = (a, n, t)
    return bad

if __name__ == "__main__":
    src = sys.argv[1]
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", *sys.argv[2:], src, "-o", f.name],
                       check=True, stderr=subprocess.DEVNULL)
        asm_mfma = '"v_mfma' in open(src).read()          # MFMAs issued as asm statements (a builtin's hazards are the compiler's)
        bad = scan(f.name) if asm_mfma else []
        early = scan_early_reads(f.name) if asm_mfma else []
        sg = scan_sgpr_into_asm_vmem(f.name)
        al = scan_async_asm_loads(f.name)
        ws = scan_asm_wide_stores(f.name)
        m0 = scan_m0_after_asm(f.name)
        tr = scan_trans_into_asm_valu(f.name)
    for pn, pt, n, t in bad: print(f"{src}: line {pn}: {pt}   ->   line {n}: {t}")
    for n, t, qn, qt in early: print(f"{src}: line {n}: {t}   read early by   line {qn}: {qt}")
    for pn, pt, n, t, st in sg: print(f"{src}: line {pn}: {pt}   ->   asm line {n}: {t}   ({st} wait states, 5 needed)")
    for n, t, qn, qt in al: print(f"{src}: asm load line {n}: {t}   destination touched in front of its wait by   line {qn}: {qt}")
    for n, t, qn, qt, st in ws: print(f"{src}: asm store line {n}: {t}   data register written {st} wait state(s) later (2 needed) by   line {qn}: {qt}")
    for pn, pt, n, t in tr: print(f"{src}: line {pn}: {pt}   ->   asm line {n}: {t}   (a transcendental's result needs one wait state)")
    for wn, wt, n, t in m0: print(f"{src}: asm line {wn}: {wt}   wrote M0, read by the compiler's   line {n}: {t}")
    print(f"{src}: {len(bad)} VALU-write -> MFMA SrcA/SrcB adjacencies, {len(early)} early reads of an MFMA result, "
          f"{len(sg)} VALU-written SGPRs read early by an asm vector-memory instruction, {len(al)} touches of an in-flight asm load's destination, "
          f"{len(ws)} VALU writes of a wide asm store's data inside its 2 wait states, {len(m0)} compiler reads of an asm-written M0, "
          f"{len(tr)} transcendental results read by the next instruction inside an asm statement")
    sys.exit(1 if bad or early or sg or al or ws or m0 or tr else 0)
