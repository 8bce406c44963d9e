"""Build the remaining compile-time variants of the kernel library and time them against the shipped build on one GPU.

    python tools/ab_variants.py            # builds lib/libtriforce_hip_<name>.so for every variant, then runs
                                           # tools/tune.py (cold-cache kernel timings) once per library

Each knob's shipped value is the measured winner (profiles/r02_nsplit_sweep.json, r02_gemm_pipeline_ab.jsonl); the
variants are the losing sides, kept buildable for re-measurement on new silicon / compilers:
  q2occ1      TF_ATTN_QT2_OCC=0        two-q-tile split-KV kernel at the compiler's own 1 wave per SIMD (298 registers)
  nofunnel    TF_TREE_MASK_FUNNEL=0    128-row tree slabs read the mask bit of every key separately
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triforce_amd.build import LIB_PATH, build_variant  # noqa: E402

VARIANTS = {"q2occ1": ["TF_ATTN_QT2_OCC=0"],
            "nofunnel": ["TF_TREE_MASK_FUNNEL=0"],
            # round 3: split-KV attention load pipelines and the precision of P (tools/attn_variants_ab.py;
            # profiles/r03_attn_pipeline_ab.jsonl).  "ps0" = round 2's kernel: P rounded once to fp16.
            "ps0": ["TF_ATTN_P_SPLIT=0"],
            "deep8": ["TF_ATTN_DEEP_TILES=8", "TF_ATTN_P_SPLIT=0"],
            "ring4": ["TF_ATTN_DEEP_TILES=8", "TF_ATTN_RING_Q1=4", "TF_ATTN_RING_Q2=4", "TF_ATTN_QT2_OCC=1", "TF_ATTN_P_SPLIT=0"],
            # round 3: prefill-attention slab forms (tools/prefill_variants_ab.py)
            "pipe0": ["TF_BLOCK_PIPE=0"], "pipe1": ["TF_BLOCK_PIPE=1"],
            "noahead": ["TF_PREFILL_AHEAD=0"],
            "sgu8": ["SG_U=8"], "sgu2": ["SG_U=2"], "sgw8": ["SG_WAVES=8"],
            "q2w8": ["TF_ATTN_Q2_WAVES=8"],
            "reslate": ["TF_SG_RES_EARLY=0"],        # residual operands loaded at the GEMM's tail (tools/gemm_resid_ab.py)
            "dma0": ["TF_BLOCK_DMA=0"], "dma1": ["TF_BLOCK_DMA=1"],
            # round 4: P fed to the PV MFMA as hi + lo fp16 in the block / prefill / tree kernels too (tools/prefill_variants_ab.py),
            # the many-split in-launch attention merge off, the retrieval scorer's round-3 grid rule
            "psplitblk": ["TF_BLOCK_P_SPLIT=1"], "draftps0": ["TF_DRAFT_P_SPLIT=0"],
            "bigmerge": ["FUSED_MERGE_BIG_SPLITS=64"], "rscoreceil": ["TF_RSCORE_CAP_CEIL=1"],
            "sgtail0": ["SG_TAIL_BATCH=0"],          # K-loop tail one chunk at a time (round 3)
            "sgprol0": ["SG_PROLOGUE_ORDER=0"],      # norm-GEMM prologue in round 3's load order (weights first, x after the fold)
            "sgprol1": ["SG_PROLOGUE_ORDER=1"],      # ... in round 4's first form (same order behind branches: the compiler threads it)
            "lnpre": ["SG_LN_PRE=1"],                # the first batch's norm weights prefetched with the prologue (+16 registers)
            "epilate0": ["SG_EPI_LATE=0"],           # plain GEMMs fetch their epilogue operands in front of the first weight batch
            "nopreload": ["!kernarg-preload"],       # built without -mllvm -amdgpu-kernarg-preload-count=14
            "ring4ps1": ["TF_ATTN_DEEP_TILES=8", "TF_ATTN_RING_Q1=4", "TF_ATTN_RING_Q2=4", "TF_ATTN_QT2_OCC=1", "TF_ATTN_P_SPLIT=1"],
            # round 6: the decode attention's K / V tiles loaded as MFMA fragments straight from memory (rounds 1-5: 16 rows x 64 bytes
            # per instruction, 5.9 TB/s) instead of full rows through a wave-private LDS tile (6.6 TB/s); and the fully two-deep
            # load loop forced onto the long streams (tools/verify_bench.py <tag> with TRIFORCE_HIP_LIB set)
            "fragloads": ["TF_ATTN_ROW_LOADS=0"],
            "eagerall": ["TF_ATTN_EAGER_TILES=1000000"]}

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
