"""Packing of resolved configs into the flat fp64 records the C-ABI takes
(layout documented in csrc/mpc_pack.h and include/mpcbatch.h)."""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np

NPARAM = 72


def pack_params(cfg: Dict) -> np.ndarray:
    """One instance: resolved config (config.resolve_config) -> NPARAM doubles."""
    p = np.zeros(NPARAM, dtype=np.float64)
    p[0] = cfg["dt"]; p[1] = cfg["tol"]; p[2] = cfg["qp_tol"]; p[3] = cfg["w_u"]; p[4] = cfg["w_qddot"]
    p[5] = cfg["px_ref"]; p[6] = cfg["vy_ref"]; p[7] = float(cfg.get("plant_integrator", 0))
    p[8:14] = cfg["wcv"]; p[14:20] = cfg["q0"]; p[20:26] = cfg["qdot0"]
    p[26:32] = cfg["qmin"]; p[32:38] = cfg["qmax"]; p[38:44] = cfg["umin"]; p[44:50] = cfg["umax"]
    p[50:56] = cfg["coeffs"]; p[56:61] = cfg["w_task"]
    p[61] = cfg.get("tol_eq", 0.0); p[62] = cfg.get("tol_ineq", 0.0); p[63] = cfg.get("tol_comp", 0.0)
    p[64] = cfg.get("levenberg_marquardt", 0.0)
    p[65] = cfg["N"]    # per-simulation horizon: lets one launch hold simulations of different horizons (ragged bucket)
    p[66] = 0.0 if cfg.get("qp_fast_path", 1) else 1.0   # 1: fast path of the QP solve OFF
    return p


_SCALARS = ((0, "dt"), (1, "tol"), (2, "qp_tol"), (3, "w_u"), (4, "w_qddot"), (5, "px_ref"), (6, "vy_ref"))
_VECTORS = ((8, "wcv", 6), (14, "q0", 6), (20, "qdot0", 6), (26, "qmin", 6), (32, "qmax", 6), (38, "umin", 6), (44, "umax", 6),
            (50, "coeffs", 6), (56, "w_task", 5))
_OPTIONAL = ((7, "plant_integrator"), (61, "tol_eq"), (62, "tol_ineq"), (63, "tol_comp"), (64, "levenberg_marquardt"))


def pack_batch(cfgs: Sequence[Dict]) -> np.ndarray:
    """All instances of a launch -> [batch, NPARAM] (column-wise: one numpy conversion per field, not per simulation)."""
    B = len(cfgs)
    p = np.zeros((B, NPARAM), dtype=np.float64)
    for col, key in _SCALARS:
        p[:, col] = [c[key] for c in cfgs]
    for col, key, n in _VECTORS:
        p[:, col:col + n] = np.asarray([c[key] for c in cfgs], dtype=np.float64).reshape(B, n)
    for col, key in _OPTIONAL:
        p[:, col] = [c.get(key, 0.0) for c in cfgs]
    p[:, 65] = [c["N"] for c in cfgs]
    p[:, 66] = [0.0 if c.get("qp_fast_path", 1) else 1.0 for c in cfgs]
    return p


# SQP_RTI buckets that differ only in the prediction horizon are merged into one RAGGED launch of the throughput engine once together
# they reach this many simulations per rank.  Re-derived at the end of round 4 with the fast path of the QP solve in both engines and the
# throughput engine's item-parallel first pass (profiles/r04_config2_shares.txt, BASELINE configs[2] per-GPU shares, one launch / one
# launch per horizon, run phase): 512 simulations 0.309 / 0.199 s, 1024: 0.322 / 0.318 s, 2048: 0.418 / 0.464 s, 4096: 0.603 / 0.848 s --
# the merge pays from ~1500 simulations on (3072 before the item-parallel pass, 1280 in round 3).
RAGGED_MIN_BATCH = 2048
SOLVER_RTI = 1


def bucket_key(cfg: Dict, ragged: bool = False):
    """Instances that can share one launch: same steps/solver options and robot, and the same horizon -- unless
    `ragged` (throughput engine: the horizon is a per-simulation parameter, slot [65] of the record)."""
    return (cfg["robot_name"], cfg.get("urdf_path"), cfg.get("ee_frame"), tuple(np.asarray(cfg["t_ee"]).tolist()),
            None if ragged else cfg["N"], cfg["Nsim"], cfg["solver_type"], cfg["max_iter"], cfg["qp_iter_max"],
            bool(cfg["fixed_step"]), int(cfg.get("precision", 0)))
