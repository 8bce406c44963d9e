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
