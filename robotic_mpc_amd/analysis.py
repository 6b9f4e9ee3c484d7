"""Post-run analysis on the logged trajectories -- numpy mirror of simulator.py:265-547.

Every formula follows the reference line by line, including its quirks (SURVEY.md
Appendix C): the task rotation is the identity, e5 uses ``R @ (v + (w . t_w))`` (a dot
product broadcast onto the linear velocity, no transpose, simulator.py:317) while the OCP
itself uses ``R^T (v + w x t_w)``.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

TASK_WEIGHT = 50.0  # mpc.w_origin_task ... w_fixed_vy_task (trajectory_optimizer.py:44-48)


def surface_value(coeffs, x, y):
    """S(x,y) (surface.py:21)."""
    a, b, c, d, e, f = coeffs
    return a * x * x + b * y * y + c * x * y + d * x + e * y + f


def surface_normal(coeffs, x, y):
    """Unit normal (S_x, S_y, -1)/|.| (surface.py:243-262); returns array (3, ...)."""
    a, b, c, d, e, f = coeffs
    nx = 2 * a * x + c * y + d
    ny = 2 * b * y + c * x + e
    nz = -np.ones_like(nx)
    nn = np.sqrt(nx * nx + ny * ny + nz * nz)
    return np.stack([nx / nn, ny / nn, nz / nn])


def compute_errors(ee_pose: np.ndarray, ee_vel: np.ndarray, coeffs, t_ee, px_ref: float, vy_ref: float) -> Dict[str, np.ndarray]:
    """simulator.py:265-344.  ee_pose (12,T) = [p; R row-major], ee_vel (6,T) = J_world qdot."""
    p = ee_pose[:3]                              # :271
    R = ee_pose[3:12].T.reshape(-1, 3, 3)        # :303, one 3x3 per step
    t = np.asarray(t_ee, dtype=np.float64).reshape(3)
    v = ee_vel[:3].T                             # :305
    w = ee_vel[3:6].T                            # :306
    tw = R @ t                                   # :309 translation_w
    p_t = p.T + tw                               # :314
    dot = np.einsum("ti,ti->t", w, tw)           # w_ee_w.T @ translation_w  (scalar per step)
    v_task = np.einsum("tij,tj->ti", R, v + dot[:, None])  # :317 (R_ee_t = I, :290-294,312)
    R_task_y = R[:, :, 1]                        # :320
    R_task_z = R[:, :, 2]                        # :321
    px, py, pz = p_t[:, 0], p_t[:, 1], p_t[:, 2]
    n = surface_normal(coeffs, px, py).T         # :327
    z_surf = surface_value(coeffs, px, py)       # :330
    g1 = z_surf - pz                             # :331
    g2 = np.einsum("ti,ti->t", n, R_task_z)      # :332
    g3 = R_task_y[:, 0]                          # :333
    g4 = px                                      # :334
    g5 = v_task[:, 1]                            # :335
    return {
        "e1": g1, "e2": 1.0 - g2, "e3": g3, "e4": px_ref - g4, "e5": vy_ref - g5,  # :337-341
        "p_task_z": pz, "p_ee_y": p[1].copy(),                                     # :342,344
    }


def compute_metrics(errors: Dict[str, np.ndarray], dt: float) -> Dict:
    """simulator.py:347-390: ITSE, RMSE, weighted RMSE (weights 50 each)."""
    itse, rmse = {}, {}
    for key in ("e1", "e2", "e3", "e4", "e5"):
        e = errors[key]
        tvec = np.arange(len(e)) * dt                       # :367
        itse[key] = float(np.sum(tvec * e ** 2) * dt)       # :368-369
        rmse[key] = float(np.sqrt(np.mean(e ** 2)))         # :375-379
    w = TASK_WEIGHT
    weighted = float(np.sqrt(np.mean(sum(w * errors[k] ** 2 for k in ("e1", "e2", "e3", "e4", "e5")))))  # :384
    return {"weighted_rmse": weighted, "rmse": rmse, "itse": itse}


def compute_solver_stats(sqp_iter, status, residuals, cost, solver_time) -> Dict:
    """simulator.py:392-420."""
    kkt = np.max(residuals, axis=1)
    return {
        "sqp_iterations": sqp_iter, "solver_status": status, "residuals": residuals, "kkt_residuals": kkt,
        "res_stat": residuals[:, 0], "res_eq": residuals[:, 1], "res_ineq": residuals[:, 2], "res_comp": residuals[:, 3],
        "cost_history": cost,
        "total_sqp_iterations": int(np.sum(sqp_iter)), "avg_sqp_iterations": float(np.mean(sqp_iter)),
        "num_failures": int(np.sum(status != 0)), "total_solver_time": float(np.sum(solver_time)),
        "max_kkt_residual": float(np.max(kkt)),
    }


def compute_timings(mpc_time, integration_time, solver_time) -> Dict:
    """simulator.py:422-448."""
    total = mpc_time + integration_time
    return {
        "mpc_time": mpc_time, "integration_time": integration_time, "solver_time": solver_time,
        "total_computation_time": total,
        "avg_mpc_time": float(np.mean(mpc_time)), "avg_solver_time": float(np.mean(solver_time)),
        "avg_integration_time": float(np.mean(integration_time)), "avg_total_time": float(np.mean(total)),
        "computational_time_sim": float(np.sum(total)),
    }


def compute_data(z, u, ee_pose, wcv, dt, px_ref, vy_ref, N) -> Dict:
    """simulator.py:450-488."""
    q, qdot = z[:6], z[6:]
    w = np.asarray(wcv, dtype=np.float64).reshape(-1, 1)
    qddot = -w * qdot + w * u                               # :466
    qddot_fd = np.empty_like(qdot)
    qddot_fd[:, :-1] = (qdot[:, 1:] - qdot[:, :-1]) / dt    # :471
    qddot_fd[:, -1] = qddot_fd[:, -2]                       # :474
    return {"time": np.arange(q.shape[1]) * dt, "q": q, "qdot": qdot, "qddot": qddot, "qddot_fd": qddot_fd, "u": u,
            "ee_pose": ee_pose, "px_ref": px_ref, "vy_ref": vy_ref, "N": N}


ERROR_ROWS = ("e1", "e2", "e3", "e4", "e5", "p_task_z", "p_ee_y")


def errors_rows(errors: Dict[str, np.ndarray]) -> np.ndarray:
    """The dict of simulator.py:337-344 as the [7, T1] array of mpcb_result.errors."""
    return np.stack([np.asarray(errors[k]) for k in ERROR_ROWS])


def errors_dict(rows: np.ndarray) -> Dict[str, np.ndarray]:
    return {k: rows[i] for i, k in enumerate(ERROR_ROWS)}


def batch_summary(errors, sqp_iter, qp_iter, status, residuals, solver_time, plant_time, dt, w_task=None) -> np.ndarray:
    """Vectorised Simulator.metrics / solver_stats / timings / get_summary (simulator.py:347-448, 509-547) for a
    whole bucket at once: arrays [B, ...] -> [B, 24] with the column layout of mpcb_summary (include/mpcbatch.h).
    The host mirror of the device summary kernel (used when a runner delivers no summary, and to check the kernel)."""
    e = np.asarray(errors)[:, :5, :]                       # [B,5,T1]
    B, _, T1 = e.shape
    dt = np.broadcast_to(np.asarray(dt, dtype=np.float64), (B,))
    w = np.full((B, 5), TASK_WEIGHT) if w_task is None else np.broadcast_to(np.asarray(w_task, dtype=np.float64), (B, 5))
    e2 = e * e
    t = np.arange(T1)[None, None, :] * dt[:, None, None]   # :367
    out = np.zeros((B, 24))
    out[:, 0:5] = np.sqrt(e2.mean(axis=2))                 # :375-379
    out[:, 5:10] = (t * e2).sum(axis=2) * dt[:, None]      # :368-369
    out[:, 10] = np.sqrt((w[:, :, None] * e2).sum(axis=1).mean(axis=1))   # :383-384
    S = np.asarray(sqp_iter).shape[1]
    out[:, 11] = np.asarray(sqp_iter).sum(axis=1)
    out[:, 12] = out[:, 11] / S
    out[:, 13] = (np.asarray(status) != 0).sum(axis=1)
    out[:, 14] = np.asarray(residuals).max(axis=(1, 2))
    tsol, tpl = np.asarray(solver_time).sum(axis=1), np.asarray(plant_time).sum(axis=1)
    out[:, 15] = tsol
    out[:, 16] = tsol / S
    out[:, 17] = tsol / S
    out[:, 18] = tpl / S
    out[:, 19] = tsol + tpl
    out[:, 20] = np.asarray(qp_iter).sum(axis=1)
    return out
