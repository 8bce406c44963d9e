"""A/B of the split-KV attention's load pipelines over prebuilt library variants (tools/ab_variants.py VARIANTS; build
them HERE — hipcc cross-compiles — so the GPU box only measures):

    python tools/attn_variants_ab.py build nodeep deep8 ring4q2 ring4 ring3      (no GPU needed)
    python tools/attn_variants_ab.py run   nodeep deep8 ring4q2 ring4 ring3 [--stages nodeep,deep8,ring4]

run: per variant tools/attn_merge_ab.py --fused-only (cold-cache hipGraph chains at the decode shapes), then — for the
variants named by --stages — tools/verify_bench.py (stage latencies of the 7B decode loop's model calls, in situ).
Output: JSON lines on stdout."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triforce_amd.build import LIB_DIR, build_variant  # noqa: E402
from tools.ab_variants import VARIANTS  # noqa: E402


def lib_of(name):
    return os.path.join(LIB_DIR, f"libtriforce_hip_{name}.so")


def main():
    mode, rest = sys.argv[1], sys.argv[2:]
    stages = []
    if "--stages" in rest:
        i = rest.index("--stages")
        stages = rest[i + 1].split(",")
        rest = rest[:i] + rest[i + 2:]
    if mode == "build":
        for n in rest:
            print(build_variant(n, VARIANTS[n], verbose=False), flush=True)
        return
    for n in rest:
        env = dict(os.environ, TRIFORCE_HIP_LIB=lib_of(n))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "attn_merge_ab.py"), "--fused-only", n], env=env,
                             capture_output=True, text=True)
        sys.stdout.write("".join(l + "\n" for l in out.stdout.splitlines() if l.startswith("{")))
        if out.returncode:
            print(json.dumps({"lib": n, "failed": out.stderr[-600:]}))
        sys.stdout.flush()
    for n in stages:
        env = dict(os.environ, TRIFORCE_HIP_LIB=lib_of(n))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "verify_bench.py"), n], env=env, capture_output=True,
                             text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(line[-1] if line else json.dumps({"tag": n, "failed": out.stderr[-600:]}), flush=True)


if __name__ == "__main__":
    main()
