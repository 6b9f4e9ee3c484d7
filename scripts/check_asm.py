"""Scan the gfx950 assembly of the rollout kernel for a known hipcc 7.2 miscompile.

A lane-divergent loop (`s_andn2_b64 exec ... s_cbranch_execnz`) falls through with an EMPTY exec
mask; hipcc 7.2 sometimes places VGPR<-AGPR spill reloads (`v_accvgpr_read`) in that fall-through
block, before exec is restored, so the reload silently does nothing.  All loop bounds in the
engine are made wave-uniform (Ex::uni) so that such loops do not exist; this script verifies it.

usage: python scripts/check_asm.py            (exit code 1 when the pattern is found)
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "robotic_mpc_amd", "csrc", "mpc_kernel.hip")


def scan(asm_text):
    lines = asm_text.split("\n")
    func, hits = None, []
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            func = m.group(1)
        if "s_cbranch_execnz" not in l:
            continue
        for j in range(i + 1, min(i + 80, len(lines))):
            t = lines[j].strip()
            if (t.startswith(".LBB") or t.startswith("s_or_b64 exec") or t.startswith("s_mov_b64 exec")
                    or t.startswith("s_or_saveexec") or "s_branch" in t or "s_cbranch" in t or "s_setpc" in t):
                break
            if t.startswith("v_") and not t.startswith(("v_readlane", "v_readfirstlane", "v_cmp")):
                hits.append((func, i + 1, j + 1, t))
                break
    return hits


# Second check (round 3): the item-parallel passes keep up to ~100 operands per lane in flight; a build whose register budget
# they exceed (the 8-wavefront geometry has 256) spills them to scratch memory, and every scratch reload waits with vmcnt(0) for
# ALL prefetches in flight -- measured: the final forward sweep with 38 scratch operations cost 12 us per MPC step (-4 %).
HOT = ("fwd_resident", "corr_resident", "residual_direct", "fact_pass_t")
MAX_SCRATCH_OPS = 8      # (callee-saved registers at a pass's entry / exit)


def scratch_ops(asm_text):
    lines = asm_text.split("\n")
    out, func = {}, None
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            func = m.group(1)
        elif l.startswith(".Lfunc_end"):
            func = None
        elif func and "scratch_" in l and any(h in func for h in HOT) and ("DevExecILi8ELi1E" in func or "DevExecILi4ELi1E" in func):
            out[func] = out.get(func, 0) + 1
    return out


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--offload-device-only",
                               "-o", out, SRC] + sys.argv[1:], cwd=d, stderr=subprocess.DEVNULL)
        text = open(out).read()
        hits = scan(text)
        spills = {f: n for f, n in scratch_ops(text).items() if n > MAX_SCRATCH_OPS}
    for h in hits:
        print("vector op under empty exec after divergent loop: %s line %d -> %d: %s" % h)
    for f, n in spills.items():
        print("register spills in a hot pass: %s: %d scratch operations" % (f, n))
    print("check_asm: %d suspicious site(s), %d hot pass(es) spilling" % (len(hits), len(spills)))
    return 1 if hits or spills else 0


if __name__ == "__main__":
    sys.exit(main())
