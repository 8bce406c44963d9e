"""ISA lint for the gfx950 kernels (no GPU needed: hipcc cross-compiles).

Hazard checked: a VALU instruction inside INLINE ASM reading a VGPR that an MFMA wrote too recently.  The compiler's hazard
recogniser inserts the wait states an MFMA result needs (11 after an 8-pass v_mfma_f32_16x16x32_f16) in front of its own
instructions, not in front of inline asm; a round-3 build of the prefill kernel returned NaN exactly this way
(csrc/attn.hip, mfma_settle8).  For every asm VALU instruction the lint walks back to the MFMA that produced each source
register (within one basic block) and adds up a LOWER bound of the cycles in between: 8 per MFMA (it holds the issue port
for its passes), N + 1 per `s_nop N`, 1 per anything else.  Fewer than 11 is reported.

    python tools/isa_lint.py [file.hip ...]      exit code 1 on findings
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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


def compile_to_asm(src, defines=()):
    hipcc = "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only"] + \
            [f"-D{x}" for x in defines] + ["-o", out, src]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def main():
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
