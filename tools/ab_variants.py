"""Build the compile-time variants that are gated OFF in the shipped library and time them against it on one GPU.

    python tools/ab_variants.py            # builds lib/libtriforce_hip_<name>.so for every variant, then runs
                                           # tools/tune.py (cold-cache kernel timings) once per library

Variants (each verified here only as far as a CPU box can: the default build's ISA is unchanged by the gates, the
variant compiles without new spills where stated, host-testable logic is tested in tests/test_native_cpu.py):
  occ3        TF_ATTN_LATE_VT=1 TF_ATTN_OCC=3   split-KV decode attention at 3 waves per SIMD: each V fragment is
                                                transposed right before its PV MFMA, 212 -> 153 registers, no spill
  funnel      TF_TREE_MASK_FUNNEL=1             128-row tree slabs read a row's 8 mask bits with one 64-bit funnel shift
                                                (2 loads instead of 8 per q-tile and slab)
  depth3      TF_ATTN_DEPTH=3                   three K/V tiles in flight per wave (measured: no gain, kept for reference)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triforce_amd.build import LIB_PATH, build_variant  # noqa: E402

VARIANTS = {"occ3": ["TF_ATTN_LATE_VT=1", "TF_ATTN_OCC=3"], "latevt": ["TF_ATTN_LATE_VT=1"],
            "funnel": ["TF_TREE_MASK_FUNNEL=1"], "depth3": ["TF_ATTN_DEPTH=3"]}

if __name__ == "__main__":
    names = sys.argv[1:] or list(VARIANTS)
    libs = {"default": LIB_PATH}
    for n in names:
        libs[n] = build_variant(n, VARIANTS[n], verbose=False)
    for tag, lib in libs.items():
        env = dict(os.environ, TRIFORCE_HIP_LIB=lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tune.py"), tag], env=env, capture_output=True,
                             text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(json.dumps({"tag": tag, "result": json.loads(line[-1]) if line else out.stderr[-400:]}), flush=True)
