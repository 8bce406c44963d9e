"""Host-side checks of header-only pieces of the kernels (plain C++ compiled with g++; no GPU)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_tree_mask_funnel_matches_the_per_key_rule(tmp_path):
    """tf_tree_vis8 (csrc/tree_mask.h, the TF_TREE_MASK_FUNNEL form of the tree-attention mask read) against the
    per-key rule, exhaustively over tree offsets / alignments / lengths and random mask rows."""
    exe = tmp_path / "test_tree_mask"
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "native", "test_tree_mask.cpp"), "-o", str(exe)],
                   check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert out.startswith("OK "), out
