"""Comparer for trajectories dumped from the REFERENCE (acados) path -- the hook that pins parity once somebody has acados.

The reference cannot run in the build pipeline (acados / CasADi / Pinocchio are absent), so nothing in
this repository pins the engine against acados itself.  A user who HAS the reference stack can dump one or more runs
with the snippet in INTEGRATION.md ("Dumping a reference run"):

    np.savez("ref_run.npz", config=json.dumps(cfg_jsonable), **sim.get_data(), **{k: sim.errors[k] for k in ("e1","e2","e3","e4","e5")},
             solver_status=sim.solver_status, sqp_iter=sim.sqp_iter)

and check them here (needs an MI355X):

    python -m robotic_mpc_amd.compare ref_run.npz [more.npz ...] [--tol 1e-6]

For every file the same config is run on the HIP engine and q, qdot, u, e1..e5 are compared sample by sample
(north_star's bound: 1e-6).  Steps at which either side reports a solver status other than 0 are path-dependent
 and end the strict comparison of that file; the report says where.
"""
from __future__ import annotations

import json
import sys
from typing import Dict, Optional

import numpy as np

KEYS = ("q", "qdot", "u", "e1", "e2", "e3", "e4", "e5")


def compare_arrays(ref: Dict[str, np.ndarray], got: Dict[str, np.ndarray], tol: float = 1e-6,
                   ref_status: Optional[np.ndarray] = None, got_status: Optional[np.ndarray] = None) -> Dict:
    """Max |ref - got| per key over the columns before the first flagged step; `ok` if all are within `tol`."""
    n_cols = min(np.asarray(ref[k]).shape[-1] for k in KEYS if k in ref)
    stop = n_cols
    for st in (ref_status, got_status):
        if st is not None:
            bad = np.nonzero(np.asarray(st) != 0)[0]
            if bad.size:
                stop = min(stop, int(bad[0]) + 1)      # column i+1 is the state after step i
    out = {"columns_compared": stop, "columns_total": n_cols, "max_abs_diff": {}, "first_violation": {}}
    ok = True
    for k in KEYS:
        if k not in ref:
            continue
        a, b = np.asarray(ref[k], dtype=np.float64)[..., :stop], np.asarray(got[k], dtype=np.float64)[..., :stop]
        if a.shape != b.shape:
            raise ValueError(f"{k}: reference shape {a.shape} vs engine shape {b.shape}")
        d = np.abs(a - b)
        out["max_abs_diff"][k] = float(d.max()) if d.size else 0.0
        if d.size and d.max() > tol:
            ok = False
            out["first_violation"][k] = int(np.nonzero(d.reshape(-1, d.shape[-1]).max(axis=0) > tol)[0][0])
    out["ok"] = ok
    out["tol"] = tol
    return out


def _config_from_npz(f) -> Dict:
    cfg = json.loads(str(f["config"]))
    for k, v in list(cfg.items()):
        if isinstance(v, list):
            cfg[k] = np.asarray(v, dtype=np.float64) if k not in ("surface_limits",) else tuple(map(tuple, v))
    return cfg


def compare_file(path: str, tol: float = 1e-6, engine=None) -> Dict:
    """Run the config stored in `path` on the HIP engine and compare with the stored reference trajectories."""
    from .simulator import Simulator

    with np.load(path, allow_pickle=False) as f:
        cfg = _config_from_npz(f)
        ref = {k: f[k] for k in KEYS if k in f.files}
        ref_status = f["solver_status"] if "solver_status" in f.files else None
    cfg["scene"] = False
    sim = Simulator(**cfg).run(engine)
    d = sim.get_data()
    got = {"q": d["q"], "qdot": d["qdot"], "u": d["u"], **{k: sim.errors[k] for k in ("e1", "e2", "e3", "e4", "e5")}}
    rep = compare_arrays(ref, got, tol, ref_status, sim.solver_status)
    rep["file"] = path
    return rep


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    tol = 1e-6
    if "--tol" in argv:
        i = argv.index("--tol")
        tol = float(argv[i + 1])
        del argv[i:i + 2]
    if not argv:
        print(__doc__)
        return 2
    bad = 0
    for path in argv:
        rep = compare_file(path, tol)
        print(json.dumps(rep))
        bad += 0 if rep["ok"] else 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
