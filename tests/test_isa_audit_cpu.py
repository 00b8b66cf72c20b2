"""The miscompile behind the round-1 maskconv_cl staging fault must not reappear in any kernel of the library: hipcc
evaluated a wave-uniform runtime flag with a VALU compare inside one divergent-exit loop and re-used the resulting lane mask
inside a SIBLING loop (tools/micro/convflag/).  tools/isa_lanemask_audit.py looks for that signature in the gfx950
assembly of every csrc/*.hip (hipcc cross-compiles here, no GPU needed)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_lanemask_audit", os.path.join(ROOT, "tools", "isa_lanemask_audit.py"))
audit = importlib.util.module_from_spec(spec)
spec.loader.exec_module(audit)

# the shape of the interim build's code (registers and labels as hipcc emitted them, bodies trimmed)
BAD = """
kernel_with_the_fault:
.LBB4_10:                               ; =>This Loop Header: Depth=1
	s_and_saveexec_b64 s[22:23], s[0:1]
.LBB4_13:                               ;   Parent Loop BB4_10 Depth=1
                                        ; =>  This Inner Loop Header: Depth=2
	v_cndmask_b32_e64 v36, 0, 1, s[48:49]
	v_cmp_ne_u32_e64 s[2:3], 1, v36
	s_and_b64 vcc, exec, s[2:3]
	s_cbranch_vccnz .LBB4_16
	s_andn2_b64 exec, exec, s[72:73]
	s_cbranch_execz .LBB4_18
.LBB4_18:                               ;   in Loop: Header=BB4_10 Depth=1
	s_or_b64 exec, exec, s[72:73]
.LBB4_20:                               ;   Parent Loop BB4_10 Depth=1
                                        ; =>  This Inner Loop Header: Depth=2
	v_cmp_gt_i32_e32 vcc, s45, v34
; %bb.21:                               ;   in Loop: Header=BB4_20 Depth=2
	s_and_b64 vcc, exec, s[2:3]
	s_cbranch_vccnz .LBB4_23
"""
GOOD = BAD.replace("	v_cmp_gt_i32_e32 vcc, s45, v34\n", "	v_cmp_gt_i32_e32 vcc, s45, v34\n	v_cmp_ne_u32_e64 s[2:3], 1, v36\n")


def test_audit_recognises_the_signature(tmp_path):
    bad, good = tmp_path / "bad.s", tmp_path / "good.s"
    bad.write_text(BAD)
    good.write_text(GOOD)
    found = audit.audit(str(bad))
    assert len(found) == 1 and found[0][3] == "s[2:3]" and found[0][4] == "BB4_13" and found[0][7] == "BB4_20"
    assert audit.audit(str(good)) == []       # recomputed inside the second loop: fine


def test_no_kernel_of_the_library_has_the_signature(tmp_path):
    import glob
    hips = sorted(glob.glob(os.path.join(audit.CSRC, "*.hip")))
    assert len(hips) >= 13
    report = {}
    for h in hips:
        found = audit.audit(audit.assemble(h, str(tmp_path)))
        if found:
            report[os.path.basename(h)] = [(f[0], f[1], f[2]) for f in found]
    assert not report, report
