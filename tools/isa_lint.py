"""ISA lint for the gfx950 kernels (no GPU needed: hipcc cross-compiles).

Hazard checked: a VALU instruction inside INLINE ASM reading a VGPR that an MFMA wrote too recently.  The compiler's hazard
recogniser inserts the wait states an MFMA result needs (11 after an 8-pass v_mfma_f32_16x16x32_f16) in front of its own
instructions, not in front of inline asm; a round-3 build of the prefill kernel returned NaN exactly this way
(csrc/attn.hip, mfma_settle8).  For every asm VALU instruction the lint walks back to the MFMA that produced each source
register (within one basic block) and adds up a LOWER bound of the cycles in between: 8 per MFMA (it holds the issue port
for its passes), N + 1 per `s_nop N`, 1 per anything else.  Fewer than 11 is reported.

    python tools/isa_lint.py [file.hip ...]      exit code 1 on findings
    python tools/isa_lint.py --prologue [file.hip ...]   per kernel: preloaded argument SGPRs, loads issued before the first wait
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NEED = 11
REG = re.compile(r"v\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def cost(inst):
    if inst.startswith("v_mfma"):
        return 8
    if inst.startswith("s_nop"):
        try:
            return int(inst.split()[1]) + 1
        except (IndexError, ValueError):
            return 1
    return 1


def lint_asm(text, window=40):
    insts, inasm, func = [], False, "?"
    for ln, line in enumerate(text.split("\n"), 1):
        t = line.strip()
        if "ASMSTART" in t:
            inasm = True
            continue
        if "ASMEND" in t:
            inasm = False
            continue
        if t.endswith(":") and not t.startswith("."):
            func = t[:-1]
        if not t or t[0] in ";.":
            continue
        if t.endswith(":"):
            insts.append((ln, "<label>", func, False))
            continue
        insts.append((ln, t, func, inasm))
    findings = []
    for i, (ln, t, func, a) in enumerate(insts):
        if not a or not t.startswith("v_"):
            continue
        ops = t.split(None, 1)[1].split(",") if " " in t else []
        srcs = set()
        for o in ops[1:]:
            srcs |= regs(o)
        gap = 0
        for back in range(1, window + 1):
            if i - back < 0:
                break
            pln, pt, _, _ = insts[i - back]
            if pt == "<label>":
                break
            if pt.startswith("v_mfma"):
                dst = regs(pt.split(None, 1)[1].split(",")[0])
                if dst & srcs:
                    if gap < NEED:
                        findings.append((func, ln, t, pln, pt, gap))
                    break
            gap += cost(pt)
            if gap >= NEED:
                break
    return findings


def compile_to_asm(src, defines=(), extra=()):
    hipcc = "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only"] + \
            [f"-D{x}" for x in defines] + list(extra) + ["-o", out, src]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


# ---- second check (round 4): the shape of a kernel's PROLOGUE ---------------------------------------------------------------
# Two things the round-4 ISA reading found at the top of the decode kernels, both invisible in the source:
#   * every kernel opened with `s_load_dwordx*; s_waitcnt lgkmcnt(0)` — the kernel-argument segment through the scalar cache
#     — before its first address; the library is built with kernarg preload and the kernels order their arguments for it;
#   * the norm-prologue GEMM's first vector-memory operation was `global_load; s_waitcnt vmcnt(0)` and the fold of the norm
#     partials ran BEFORE the first weight load was issued (the compiler sank / threaded the conditional loads).
# prologue_shape() reports, per kernel: the preloaded SGPR count, whether a scalar wait precedes the first vector load of the
# real entry (behind the backward-compatible s_load + s_branch header), the number of vector loads issued before the first
# vmcnt wait, and that wait's count (0 = everything outstanding is waited for: a serialisation when loads follow).
def prologue_shape(text):
    out, func, state, desc = {}, None, None, None
    for line in text.split("\n"):
        t = line.strip()
        m = re.match(r"^\.amdhsa_kernel\s+(\S+)", t)
        if m:
            desc = m.group(1)
            continue
        m = re.match(r"^\.amdhsa_user_sgpr_kernarg_preload_length\s+(\d+)", t)
        if m and desc is not None:
            out.setdefault(desc, {})["preload"] = int(m.group(1))
            continue
        m = re.match(r"^([A-Za-z_]\w*):\s*(;.*)?$", line)
        if m:                                               # a function label (local labels start with '.')
            func = m.group(1)
            state = dict(loads=0, scalar_wait=False, done=False)
            out.setdefault(func, {}).update(loads_before_first_wait=None, first_vmcnt=None, scalar_wait_before_first_load=None)
            continue
        if func is None or state is None or state["done"] or not t or t[0] == ";":
            continue
        if re.match(r"^\.LBB\d+_0:", t):                    # real entry behind the kernarg-preload compatibility header
            state.update(loads=0, scalar_wait=False)
            continue
        if t[0] == ".":
            continue
        if t.startswith("s_endpgm"):
            state["done"] = True
            continue
        if t.startswith("s_waitcnt") and "lgkmcnt" in t and state["loads"] == 0:
            state["scalar_wait"] = True
        if re.match(r"(global_load|buffer_load|flat_load)", t):
            if state["loads"] == 0:
                out[func]["scalar_wait_before_first_load"] = state["scalar_wait"]
            state["loads"] += 1
        m = re.search(r"vmcnt\((\d+)\)", t) if t.startswith("s_waitcnt") else None
        if m and state["loads"] > 0:
            out[func]["loads_before_first_wait"] = state["loads"]
            out[func]["first_vmcnt"] = int(m.group(1))
            state["done"] = True
    return out




def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--prologue":           # table of prologue shapes, built like the library
        from triforce_amd.build import FLAGS
        extra = [f for f in FLAGS if f in ("-mllvm",) or f.startswith("-amdgpu-")]
        for f in sys.argv[2:] or [os.path.join(ROOT, "triforce_amd", "csrc", "gemv.hip")]:
            for func, sh in sorted(prologue_shape(compile_to_asm(f, extra=extra)).items()):
                if sh.get("loads_before_first_wait") is not None:
                    print(f"{func[:70]:70s} preload {sh.get('preload')}  scalar wait first: {sh['scalar_wait_before_first_load']}  "
                          f"loads before first vmcnt wait: {sh['loads_before_first_wait']}  vmcnt({sh['first_vmcnt']})")
        return 0
    files = sys.argv[1:] or [os.path.join(ROOT, "triforce_amd", "csrc", f) for f in ("attn.hip", "gemv.hip", "sampling.hip")]
    bad = 0
    for f in files:
        for func, ln, t, pln, pt, gap in lint_asm(compile_to_asm(f)):
            bad += 1
            print(f"{os.path.basename(f)}: {func[:60]}: asm `{t}` (line {ln}) reads the result of `{pt[:50]}` (line {pln}) "
                  f"after >= {gap} cycles, needs {NEED}")
    print(f"isa_lint: {bad} finding(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())


# ---- third check (round 6): the in-launch hand-off kernels (csrc/draft_persist.hip, csrc/topp_multi.hip) ----------------------
# Their correctness rests on instruction-level facts the source only implies (MI355X_MICROARCH.md, inter-workgroup visibility):
#   * data handed to another workgroup is stored write-through and read past the L1: `global_store* ... sc1` / `global_load* ... sc1`;
#   * every arrival (a returning `global_atomic_add ... sc1` on a counter) is preceded by `s_waitcnt vmcnt(0)` — the drain of the
#     stores it publishes — with no store in between;
#   * there is no agent-scope FENCE in the kernels (`buffer_wbl2` / `buffer_inv`): a fence that crept in (e.g. a seq_cst atomic)
#     would cost ~1.7 us per edge and hide a missing drain.
# handoff_shape() reports, per kernel: sc1 loads / stores, fences, arrivals and how many of them lack the drain, registers, spills.
def handoff_shape(text):
    out = {}
    for m in re.finditer(r"^(_Z\w+):\s*;\s*@\1\n([\s\S]*?)\n\.Lfunc_end\d+:", text, re.M):
        name, body = m.group(1), m.group(2)
        lines = [l.strip() for l in body.split("\n")]
        arrivals = undrained = 0
        drained = False                                   # a vmcnt(0) wait with no store issued since
        for l in lines:
            if l.startswith("s_waitcnt") and re.search(r"vmcnt\(0\)", l):
                drained = True
            elif re.match(r"(global|flat|buffer)_store", l):
                drained = False
            elif re.match(r"global_atomic_add(_u32)?\s+v\d+", l) and "sc0" in l:      # returning add = an arrival
                arrivals += 1
                undrained += 0 if drained else 1
        out[name] = {"sc1_loads": sum(1 for l in lines if l.startswith("global_load") and " sc1" in l),
                     "sc1_stores": sum(1 for l in lines if l.startswith("global_store") and " sc1" in l),
                     "fences": sum(1 for l in lines if l.startswith("buffer_wbl2") or l.startswith("buffer_inv")),
                     "arrivals": arrivals, "arrivals_without_drain": undrained}
    for m in re.finditer(r"\.name:\s+(_Z\w+)[\s\S]*?\.vgpr_count:\s+(\d+)[\s\S]*?\.vgpr_spill_count:\s+(\d+)", text):
        if m.group(1) in out:
            out[m.group(1)].update(vgprs=int(m.group(2)), vgpr_spills=int(m.group(3)))
    return out
