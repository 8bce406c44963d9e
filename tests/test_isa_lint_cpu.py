"""ISA-level check that needs no GPU (hipcc cross-compiles gfx950 here): no inline-asm VALU instruction may read an MFMA
result before the wait states the hardware requires have passed — the compiler guards its own instructions, not inline asm
(tools/isa_lint.py; found in round 3 when a variant build of the prefill kernel returned NaN)."""
import os
import shutil

import pytest

from tools import isa_lint

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "triforce_amd", "csrc")
pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                                reason="hipcc not available")


def test_lint_catches_the_round3_hazard():
    bad = """
_Zkernel:
	v_mfma_f32_16x16x32_f16 v[158:161], v[146:149], v[10:13], v[138:141]
	;;#ASMSTART
	v_max3_f32 v234, v158, v159, v160
	;;#ASMEND
	s_endpgm
"""
    assert len(isa_lint.lint_asm(bad)) == 1
    ok = bad.replace("\t;;#ASMSTART\n\tv_max3", "\ts_nop 7\n\ts_nop 3\n\t;;#ASMSTART\n\tv_max3")
    assert isa_lint.lint_asm(ok) == []


@pytest.mark.parametrize("defines", [(), ("TF_BLOCK_PIPE=0",), ("TF_BLOCK_PIPE=1",), ("TF_ATTN_P_SPLIT=0",), ("TF_PREFILL_AHEAD=0",), ("TF_BLOCK_DMA=0",)])
def test_attention_kernels_have_no_unguarded_asm_read_of_an_mfma_result(defines):
    text = isa_lint.compile_to_asm(os.path.join(CSRC, "attn.hip"), defines)
    findings = isa_lint.lint_asm(text)
    assert findings == [], findings[:3]


def test_norm_gemm_prologue_issues_every_load_before_its_first_wait():
    """Round 4: built with the library's flags, the norm-prologue forms of skinny_gemm_kernel must (a) get
    their 14 leading argument dwords preloaded into SGPRs, (b) issue their first vector load without a scalar wait, and
    (c) issue the norm partials, the x rows and the first weights — >= 24 loads — before the first vmcnt wait, which must
    leave the later loads in flight (a count > 0).  Round 4's first build failed all three: `s_load x6; s_waitcnt
    lgkmcnt(0)` at entry, then `global_load; s_waitcnt vmcnt(0); v_add`, the fold in front of the first weight load —
    none of it visible in the source.  The synthetic text pins the parser."""
    old = """
\t.amdhsa_kernel _Zk
\t\t.amdhsa_user_sgpr_kernarg_preload_length 0
_Zk: ; @_Zk
\ts_load_dwordx4 s[12:15], s[0:1], 0x0
\ts_waitcnt lgkmcnt(0)
\tglobal_load_dword v2, v2, s[26:27]
\ts_waitcnt vmcnt(0)
\tv_add_f32_e32 v23, 0, v2
\tglobal_load_dwordx4 v[4:7], v[8:9], off
\ts_endpgm
"""
    sh = isa_lint.prologue_shape(old)["_Zk"]
    assert sh == dict(preload=0, scalar_wait_before_first_load=True, loads_before_first_wait=1, first_vmcnt=0)
    from triforce_amd.build import FLAGS
    extra = [f for f in FLAGS if f == "-mllvm" or f.startswith("-amdgpu-")]
    assert extra, "the library is built with kernarg preload"
    shapes = isa_lint.prologue_shape(isa_lint.compile_to_asm(os.path.join(CSRC, "gemv.hip"), extra=extra))
    norm = {k: v for k, v in shapes.items() if k.startswith("_Z18skinny_gemm_kernelILi1E") and "ELb1ELi" in k[:40]
            and v.get("loads_before_first_wait") is not None}
    # one row tile, norm prologue, no K split across workgroups, UX = 1: q|k|v / gate|up / lm_head forms at 4 and 8 waves
    hot = {k: v for k, v in norm.items() if "ELb0ELb0ELi1EE" in k}
    assert len(hot) >= 8, sorted(norm)[:4]
    for k, v in hot.items():
        assert v.get("preload") == 14, (k, v)
        assert v["scalar_wait_before_first_load"] is False, (k, v)
        assert v["loads_before_first_wait"] >= 24 and v["first_vmcnt"] > 0, (k, v)
    # round 5: the narrow-panel forms (skinny_gemm_n8_kernel) keep the same prologue shape, and none of them may spill — at
    # <= 1 workgroup per CU they are allowed every register, but the epilogue's fully unrolled wave merge once cost 171 spills
    n8 = {k: v for k, v in shapes.items() if k.startswith("_Z21skinny_gemm_n8_kernel") and v.get("loads_before_first_wait") is not None}
    assert len(n8) >= 10, sorted(shapes)[:4]
    for k, v in n8.items():
        assert v.get("preload") == 14 and v["scalar_wait_before_first_load"] is False, (k, v)
        assert v["loads_before_first_wait"] >= 24 and v["first_vmcnt"] > 0, (k, v)
    text = isa_lint.compile_to_asm(os.path.join(CSRC, "gemv.hip"), extra=extra)
    import re
    spills = re.findall(r"\.name:\s+(_Z21skinny_gemm_n8_kernel\S+)[\s\S]*?\.vgpr_spill_count:\s+(\d+)", text)
    assert len(spills) >= 10 and all(int(n) == 0 for _, n in spills), [(k[:48], n) for k, n in spills if int(n)]


def test_handoff_kernels_store_write_through_drain_before_every_arrival_and_use_no_fences():
    """Round 6: the one-launch draft forward and the multi-workgroup top-p hand data between workgroups INSIDE a
    launch.  What makes that correct on gfx950 is visible only in the ISA: the payload goes out with `sc1` (write-through) stores
    and comes in with `sc1` loads (past this CU's L1), every arrival — a returning atomic add on an edge counter — sits behind an
    `s_waitcnt vmcnt(0)` with no store in between, and no agent-scope fence (`buffer_wbl2` / `buffer_inv`) exists at all (one that
    crept in would cost ~1.7 us per edge and could hide a missing drain).  Also pinned: no register spills where none are expected."""
    synthetic = """
_Zk: ; @_Zk
\tglobal_store_dwordx2 v[0:1], v[2:3], off sc1
\tglobal_atomic_add_u32 v4, v[5:6], v7, off sc0 sc1
\ts_waitcnt vmcnt(0)
\tglobal_atomic_add_u32 v4, v[5:6], v7, off sc0 sc1
\tglobal_load_dwordx2 v[8:9], v[10:11], off sc1
\tbuffer_inv sc1
\ts_endpgm
.Lfunc_end0:
"""
    sh = isa_lint.handoff_shape(synthetic)["_Zk"]
    assert sh == dict(sc1_loads=1, sc1_stores=1, fences=1, arrivals=2, arrivals_without_drain=1)
    from triforce_amd.build import FLAGS
    extra = [f for f in FLAGS if f == "-mllvm" or f.startswith("-amdgpu-")]
    draft = isa_lint.handoff_shape(isa_lint.compile_to_asm(os.path.join(CSRC, "draft_persist.hip"), extra=extra))
    d1 = next(v for k, v in draft.items() if "draft_persist_kernelILi1E" in k)
    d2 = next(v for k, v in draft.items() if "draft_persist_kernelILi2E" in k)
    # 5 layer edges per layer + lm + the top-p edge, some arrived at from two code paths (with / without the probability row)
    assert d1["arrivals"] >= 8 and d2["arrivals"] >= 13
    for v in (d1, d2):
        assert v["arrivals_without_drain"] == 0 and v["fences"] == 0, v
        assert v["sc1_loads"] >= 60 and v["sc1_stores"] >= 40, v
    assert d1["vgpr_spills"] == 0 and d2["vgpr_spills"] <= 48, (d1, d2)      # (two layers sit at the 256-register limit: DESIGN 15.1)
    topp = isa_lint.handoff_shape(isa_lint.compile_to_asm(os.path.join(CSRC, "topp_multi.hip"), extra=extra))
    assert len(topp) == 2
    for k, v in topp.items():
        assert v["arrivals"] == 2 and v["arrivals_without_drain"] == 0 and v["fences"] == 0, (k, v)
        assert v["sc1_loads"] >= 20 and v["sc1_stores"] >= 8 and v["vgpr_spills"] == 0 and v["vgprs"] <= 128, (k, v)
