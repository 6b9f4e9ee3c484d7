"""Instruction counts of the phases of a pass: segments of its gfx950 ISA between wave / workgroup barriers.
usage: python scripts/isa_segments.py k.s <function substring> [variant substring, default DevExecILi4ELi1E]"""
import re, sys
path, fn = sys.argv[1], sys.argv[2]
var = sys.argv[3] if len(sys.argv) > 3 else "DevExecILi4ELi1E"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and fn in l and var in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
seg, segs = {"n": 0, "valu": 0, "f64": 0, "ds": 0, "vmem": 0, "salu": 0, "start": start}, []
for i in range(start, end):
    t = lines[i].strip()
    if not t or t.startswith((";", ".", "_Z")) and "wave barrier" not in t:
        continue
    if "wave barrier" in t or t.startswith("s_barrier"):
        segs.append(seg); seg = {"n": 0, "valu": 0, "f64": 0, "ds": 0, "vmem": 0, "salu": 0, "start": i}
        continue
    op = t.split()[0]
    seg["n"] += 1
    if op.startswith("v_"): seg["valu"] += 1
    if "_f64" in op: seg["f64"] += 1
    if op.startswith("ds_"): seg["ds"] += 1
    if op.startswith(("global_", "flat_", "scratch_", "buffer_")): seg["vmem"] += 1
    if op.startswith("s_"): seg["salu"] += 1
segs.append(seg)
print(f"{fn}: {end - start} lines, {sum(s['n'] for s in segs)} instructions, {len(segs)} segments")
for k, s in enumerate(segs):
    if s["n"] >= 20:
        print(f"  seg {k:3d} line {s['start'] - start:6d}: {s['n']:5d} instr  valu {s['valu']:5d} (f64 {s['f64']:4d})  ds {s['ds']:4d}  vmem {s['vmem']:4d}  salu {s['salu']:4d}")
