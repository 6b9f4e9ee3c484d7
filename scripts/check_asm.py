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


# Second check (round 3, widened in round 4): the item-parallel passes keep up to ~100 operands per lane in flight; a build whose
# register budget they exceed (the 8-wavefront and the two-per-CU geometries have 256) spills them to scratch memory, and every
# scratch reload waits with vmcnt(0) for ALL prefetches in flight -- measured: the final forward sweep with 38 scratch operations
# cost 12 us per MPC step (-4 %).  Guarded: the hot passes of EVERY shipped latency kernel (<8,1>, <4,1>, <4,2>; <2,1> and <1,1> are
# test geometries) and the sweeps of the throughput engine, which must hold nothing in scratch beyond a pass's entry / exit
# (MAX_SCRATCH_OPS); and a recorded BUDGET for the functions that do live with scratch -- the once-per-step NLP pass (task_lin holds
# ~130 doubles), the plant log, the SQP merit pass, the kernel bodies (the inlined solver driver: engine object, residual arrays) --
# so that growth there is a decision, not an accident.  Who owns the kernels' scratch bytes (BENCH roofline.scratch_bytes): these.
HOT = ("fwd_resident", "corr_resident", "residual_direct", "fact_pass_t", "fast_rhs", "fast_commit", "forward_step_pass", "corrector_bwd_pass")
HOT_GEOM = ("DevExecILi8ELi1E", "DevExecILi4ELi1E", "DevExecILi4ELi2E")
HOT_STREAM = ("se9fact_pass", "se12forward_pass", "se14corrector_pass", "se13residual_pass", "se14residual_items", "se11fast_commit", "se12nlp_res_pass", "se9rti_items")
MAX_SCRATCH_OPS = 8      # (callee-saved registers at a pass's entry / exit)
# function-name fragment -> scratch operations allowed (the count at the commit that recorded it, + ~10 %)
BUDGET = {
    # (round 4, end: the joint-angle sincos of mpc_kin.h replaced the library's, whose large-argument path was what these functions spilled
    # around: nlp_direct 121 / 147 -> 41 / 54, log_state 43 -> 2, merit_pass 394 -> 186, the throughput engine's lin_pass 56 -> 0 and
    # log_state 33 -> 0 -- those two are held to the hot passes' limit now)
    "DevExecILi8ELi1EEE10nlp_direct": 46, "DevExecILi4ELi2EEE10nlp_direct": 60, "DevExecILi4ELi1EEE10nlp_direct": 8,
    "DevExecILi8ELi1EEE9log_state": 8, "DevExecILi4ELi2EEE9log_state": 8, "DevExecILi4ELi1EEE9log_state": 8,
    "10merit_pass": 210, "se8lin_pass": 8, "se9log_state": 8,
    "18mpc_rollout_kernelILi8ELi1E": 340, "18mpc_rollout_kernelILi4ELi2E": 330, "18mpc_rollout_kernelILi4ELi1E": 258,
    "17mpc_stream_kernelId": 160, "17mpc_stream_kernelIf": 160,
}


def scratch_by_function(asm_text):
    out, func = {}, None
    for l in asm_text.split("\n"):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            func = m.group(1)
            out.setdefault(func, 0)
        elif l.startswith(".Lfunc_end"):
            func = None
        elif func and "scratch_" in l:
            out[func] += 1
    return out


def scratch_ops(asm_text):
    """Functions over their limit: {name: (count, limit)}."""
    bad = {}
    for func, n in scratch_by_function(asm_text).items():
        limit = None
        for frag, b in BUDGET.items():
            if frag in func:
                limit = b
        if limit is None:
            hot = (any(h in func for h in HOT) and any(g in func for g in HOT_GEOM)) or any(h in func for h in HOT_STREAM)
            limit = MAX_SCRATCH_OPS if hot else None
        if limit is not None and n > limit:
            bad[func] = (n, limit)
    return bad


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--offload-device-only",
                               "-o", out, SRC] + sys.argv[1:], cwd=d, stderr=subprocess.DEVNULL)
        text = open(out).read()
        hits = scan(text)
        spills = scratch_ops(text)
    for h in hits:
        print("vector op under empty exec after divergent loop: %s line %d -> %d: %s" % h)
    for f, (n, limit) in spills.items():
        print("scratch operations over the limit: %s: %d (limit %d)" % (f, n, limit))
    print("check_asm: %d suspicious site(s), %d function(s) over their scratch limit" % (len(hits), len(spills)))
    return 1 if hits or spills else 0


if __name__ == "__main__":
    sys.exit(main())
