"""Build libtriforce_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so travels with the tree."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libtriforce_hip.so")
SOURCES = ["attn.hip", "retrieval.hip", "elementwise.hip", "sampling.hip", "offload.hip", "gemv.hip", "allreduce.hip", "draft.hip", "draft_persist.hip", "topp_multi.hip", "abi.hip"]
# -amdgpu-kernarg-preload-count: the leading pointer / scalar kernel arguments (up to 14 dwords) arrive in SGPRs with the
# dispatch instead of through s_load + s_waitcnt at the top of every kernel (gfx950 hardware feature; the compiler keeps a
# backward-compatible entry for firmware without it).  The decode kernels order their arguments for it (csrc/gemv.hip).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-gpu-rdc", "-Wno-unused-result", "-mllvm", "-amdgpu-kernarg-preload-count=14"]


def hipcc_path():
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found: cannot build libtriforce_hip.so")
    return p


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "tree_mask.h"), os.path.join(CSRC, "select_common.h"),
                        os.path.join(HERE, "..", "include", "triforce_hip.h"), os.path.abspath(__file__)]   # (flags live here)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_variant(name, defines, verbose=True):
    """Tuning builds (A/B of compile-time knobs): lib/libtriforce_hip_<name>.so, selected with TRIFORCE_HIP_LIB."""
    out = os.path.join(LIB_DIR, f"libtriforce_hip_{name}.so")
    # "NAME=VALUE" -> -DNAME=VALUE; an entry that starts with "-" is a raw compiler flag, "!-flag" removes a default flag
    drop = {d[1:] for d in defines if d.startswith("!")}
    flags = [f for f in FLAGS if f not in drop]
    if "kernarg-preload" in drop:                                # "!kernarg-preload": build without the preload flag pair
        flags = [f for f in flags if f != "-mllvm" and not f.startswith("-amdgpu-kernarg-preload-count")]
    cmd = [hipcc_path()] + flags + [d if d.startswith("-") else f"-D{d}" for d in defines if not d.startswith("!")] \
        + sources() + ["-o", out]
    if verbose:
        print("[triforce_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


def build_library(force=False, verbose=True):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc_path()] + FLAGS + sources() + ["-o", LIB_PATH]
    if verbose:
        print("[triforce_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    build_library(force=True)
