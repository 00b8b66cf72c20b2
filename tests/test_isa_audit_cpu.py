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

# beam_round_kernel (csrc/rnnt_decode.hip) as hipcc emitted it in round 4: loop BB5_54 leaves compare results in s[18:19] and
# s[22:23]; the sibling loop BB5_58 starts NEW live ranges in those registers (`implicit-def`) and keeps two loop-carried
# per-lane booleans with the merge `(old & ~exec) | (new & exec)` -- not the signature
MERGE = """
beam_round_like:
.LBB5_54:                               ; =>This Inner Loop Header: Depth=1
	v_cmp_eq_u32_e64 s[18:19], v10, v1
	v_cmp_eq_u32_e64 s[22:23], v13, v1
	s_or_b64 s[10:11], s[10:11], s[18:19]
	s_or_b64 s[8:9], s[8:9], s[22:23]
	s_or_b64 s[96:97], vcc, s[96:97]
	s_andn2_b64 exec, exec, s[96:97]
	s_cbranch_execnz .LBB5_54
; %bb.55:
	s_or_b64 exec, exec, s[96:97]
; %bb.57:
	s_mov_b64 s[20:21], 0
                                        ; implicit-def: $sgpr18_sgpr19
                                        ; implicit-def: $sgpr22_sgpr23
.LBB5_58:                               ; =>This Inner Loop Header: Depth=1
	v_cmp_eq_u32_e64 s[8:9], v4, v1
	v_cmp_eq_u32_e64 s[10:11], v5, v1
	s_or_b64 s[12:13], s[12:13], s[10:11]
	s_or_b64 s[14:15], s[14:15], s[8:9]
	s_or_b64 s[20:21], vcc, s[20:21]
	s_andn2_b64 s[6:7], s[22:23], exec
	s_and_b64 s[8:9], s[14:15], exec
	s_andn2_b64 s[10:11], s[18:19], exec
	s_and_b64 s[18:19], s[12:13], exec
	s_or_b64 s[22:23], s[6:7], s[8:9]
	s_or_b64 s[18:19], s[10:11], s[18:19]
	s_andn2_b64 exec, exec, s[20:21]
	s_cbranch_execnz .LBB5_58
"""


def test_audit_recognises_the_signature(tmp_path):
    bad, good = tmp_path / "bad.s", tmp_path / "good.s"
    bad.write_text(BAD)
    good.write_text(GOOD)
    found = audit.audit(str(bad))
    assert len(found) == 1 and found[0][3] == "s[2:3]" and found[0][4] == "BB4_13" and found[0][7] == "BB4_20"
    assert audit.audit(str(good)) == []       # recomputed inside the second loop: fine


def test_audit_does_not_report_new_live_ranges_or_the_per_lane_merge(tmp_path):
    merge = tmp_path / "merge.s"
    merge.write_text(MERGE)
    assert audit.audit(str(merge)) == []
    # each rule on its own: without the implicit-def comments the merge idiom alone clears it ...
    merge.write_text("\n".join(l for l in MERGE.splitlines() if "implicit-def" not in l))
    assert audit.audit(str(merge)) == []
    # ... and a plain read of the stale mask in the sibling loop (no merge, no new live range) is still the signature
    stale = "\n".join(l for l in MERGE.splitlines() if "implicit-def" not in l).replace(
        "s_andn2_b64 s[6:7], s[22:23], exec", "s_and_b64 s[6:7], s[22:23], s[14:15]")
    merge.write_text(stale)
    found = audit.audit(str(merge))
    assert len(found) == 1 and found[0][3] == "s[22:23]" and found[0][4] == "BB5_54" and found[0][7] == "BB5_58"
    # with the implicit-def in place the same read is of a new value: not reported
    merge.write_text(MERGE.replace("s_andn2_b64 s[6:7], s[22:23], exec", "s_and_b64 s[6:7], s[22:23], s[14:15]"))
    assert audit.audit(str(merge)) == []


# a block of an INNER loop laid out before that loop's header label reads a mask the OUTER loop made: nesting, not a sibling
NESTED = """
nested_like_beam_kernel:
.LBB0_28:                               ; =>This Loop Header: Depth=1
	v_cmp_nge_f32_e64 s[24:25], s46, v23
	s_branch .LBB0_91
.LBB0_89:                               ;   in Loop: Header=BB0_91 Depth=2
	v_cndmask_b32_e64 v79, 0, v79, s[24:25]
.LBB0_91:                               ;   Parent Loop BB0_28 Depth=1
                                        ; =>  This Inner Loop Header: Depth=2
	v_add_u32_e32 v23, 0x400, v23
	s_cbranch_execnz .LBB0_89
"""


def test_audit_knows_the_nesting_of_blocks_laid_out_before_their_loop_header(tmp_path):
    f = tmp_path / "nested.s"
    f.write_text(NESTED)
    assert audit.audit(str(f)) == []
    # the same read from a loop that is NOT inside the defining loop is still reported
    f.write_text(NESTED.replace(";   Parent Loop BB0_28 Depth=1\n                                        ; =>  This Inner Loop Header: Depth=2",
                                "; =>This Loop Header: Depth=1").replace("Header=BB0_91 Depth=2", "Header=BB0_91 Depth=1"))
    assert len(audit.audit(str(f))) == 1


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
