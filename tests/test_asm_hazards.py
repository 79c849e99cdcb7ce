"""hipcc cannot see inside inline asm: the kernels that issue their MFMAs as asm statements are checked, on their device assembly, for the
two adjacencies it would otherwise have guarded with wait states (tools/check_mfma_hazards.py) -- a VALU write (v_accvgpr_read of a weight
the compiler parked in an AGPR, v_mov) right in front of an MFMA that reads it as SrcA / SrcB, and a read of an MFMA's result before the
drain (the phi copies of a branch merge).  Both were found the hard way in round 3 (lstm_cluster16.hip: a row tile off by 1e-3 / 1e-4)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "arm-pose-estimation_amd", "csrc")


@pytest.mark.parametrize("name", ["lstm_cluster16.hip", "lstm_cluster32.hip", "lstm_upper32.hip", "lstm_cluster_f16v2.hip", "lstm_mc_small.hip",
                                  "lstm_cluster_small.hip", "lstm_upper128.hip"])
def test_no_unguarded_adjacency_around_asm_mfmas(name):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_mfma_hazards.py"), name], cwd=CSRC, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 VALU-write -> MFMA SrcA/SrcB adjacencies, 0 early reads" in r.stdout
    assert "0 VALU-written SGPRs read early by an asm vector-memory instruction" in r.stdout


def test_the_checker_sees_both_patterns(tmp_path):
    """a hand-written assembly fragment with one of each"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_mfma_hazards as chk
    f = tmp_path / "k.s"
    f.write_text("""
	v_accvgpr_read_b32 v52, a112
	v_mfma_f32_16x16x4_f32 v[118:121], v52, v124, v[118:121]
	v_mfma_f32_16x16x4_f32 v[118:121], v53, v125, v[118:121]
	s_cbranch_vccnz .LBB0_2
	s_nop 0
.LBB0_2:
	v_mov_b32_e32 v141, v121
	s_nop 15
	v_mov_b32_e32 v140, v120
	s_endpgm
""")
    assert len(chk.scan(str(f))) == 1
    early = chk.scan_early_reads(str(f))
    assert len(early) == 1 and early[0][3].startswith("v_mov_b32_e32 v141")
    # round 4: a spilled buffer descriptor reloaded by v_readlane_b32 right in front of an asm load (5 wait states needed)
    g = tmp_path / "g.s"
    g.write_text("""
	v_readlane_b32 s48, v161, 4
	v_readlane_b32 s51, v161, 7
	;;#ASMSTART
	buffer_load_dwordx4 v[74:77], v78, s[48:51], 0 offen sc1
	s_waitcnt vmcnt(0)
	;;#ASMEND
	v_readlane_b32 s52, v161, 4
	;;#ASMSTART
	s_nop 4
	buffer_load_dwordx4 v[74:77], v78, s[52:55], 0 offen sc1
	;;#ASMEND
	v_readlane_b32 s60, v161, 4
	s_nop 4
	buffer_store_dwordx2 v[96:97], v70, s[60:63], 0 offen
""")
    hits = chk.scan_sgpr_into_asm_vmem(str(g))
    assert len(hits) == 2 and all("s[48:51]" in h[3] for h in hits)      # the padded load and the compiler's own store are fine
