"""Diagnostic (timing only, results are garbage): where the factorisation sweep's stage time goes.
Builds variants of the library with a FIXED number of interior-point iterations per step and parts of the
sweep knocked out, and prints the per-section device times of each."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robotic_mpc_amd import build as b

VARIANTS = {
    "fix2": ["MPCB_PROFILE", "MPCB_DIAG_FIXED_IT=2"],
    "fix2_nores": ["MPCB_PROFILE", "MPCB_DIAG_FIXED_IT=2", "MPCB_NO_RESIDENT"],
    "fix2_nophi": ["MPCB_PROFILE", "MPCB_DIAG_FIXED_IT=2", "MPCB_DIAG_NO_PHI"],
    "fix2_novec": ["MPCB_PROFILE", "MPCB_DIAG_FIXED_IT=2", "MPCB_DIAG_NO_VEC"],
    "fix2_noldl": ["MPCB_PROFILE", "MPCB_DIAG_FIXED_IT=2", "MPCB_DIAG_NO_LDL"],
    "fix2_nob": ["MPCB_PROFILE", "MPCB_DIAG_FIXED_IT=2", "MPCB_DIAG_NO_B"],
    "fix2_noca": ["MPCB_PROFILE", "MPCB_DIAG_FIXED_IT=2", "MPCB_DIAG_NO_CA"],
    "fix2_nob_noca": ["MPCB_PROFILE", "MPCB_DIAG_FIXED_IT=2", "MPCB_DIAG_NO_B", "MPCB_DIAG_NO_CA", "MPCB_DIAG_NO_VEC"],
}
names = sys.argv[1:] or list(VARIANTS)
for n in names:
    lib = b.build_variant(n, VARIANTS[n])
    print(f"===== {n}: {' '.join(VARIANTS[n])}", flush=True)
    env = dict(os.environ, MPCB_LIB=lib)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_prof.py"), "256", "100", "1.0"], env=env,
                         capture_output=True, text=True, timeout=600)
    lines = (out.stdout + out.stderr).splitlines()
    keep = [l for l in lines if l.startswith("B=") or (l.strip().split(" ")[0] in ("fact", "bwd", "fwd", "res", "seq_fact", "seq_fwd", "seq_bwd", "io", "total", "nlp"))]
    print("\n".join(keep[:12]), flush=True)
