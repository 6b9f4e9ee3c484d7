"""On-disk format for batched results + resume of partially completed grid searches.

The reference keeps results only in memory (a list of dicts from ``SimulationManager.run_all``,
simulator.py:668-674).  One ``.npz`` archive holds a whole run, batch-major:

    names            [K]  str          queue names (simulator.py:660)
    configs          [K]  str          the Simulator(**config) dict of each simulation as JSON
    keys             [K]  str          sha1 of (name, canonical config) -- the resume key
    z [K,12,T1] u [K,6,T1] ee_pose [K,12,T1] ee_rpy [K,3,T1] ee_vel [K,6,T1] errors [K,7,T1]   (padded to the longest run)
    status/sqp_iter/qp_iter [K,T] residuals [K,T,4] cost/solver_time/plant_time [K,T]
    nsim             [K]  int          closed-loop steps of each simulation (its arrays use [:nsim(+1)])

Everything is plain numpy (no pickle), so the archive is readable without this package.
"""
from __future__ import annotations

import hashlib
import json
import os
from typing import Dict, List, Sequence

import numpy as np

LOG_KEYS_T1 = ("z", "u", "ee_pose", "ee_rpy", "ee_vel", "errors")            # [.., Nsim+1]
LOG_KEYS_T = ("status", "sqp_iter", "qp_iter", "cost", "solver_time", "plant_time")  # [Nsim]
FORMAT_VERSION = 2


def _jsonable(v):
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    if isinstance(v, dict):
        return {str(k): _jsonable(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    return v


def config_json(config: Dict) -> str:
    return json.dumps(_jsonable(config), sort_keys=True)


def resume_key(name: str, config: Dict) -> str:
    return hashlib.sha1((name + "\n" + config_json(config)).encode()).hexdigest()


def records_of(sim) -> Dict[str, np.ndarray]:
    """The raw logs of a run Simulator, as the runner delivered them."""
    sm = sim.simulation_model
    return {"z": sm.z, "u": sm.u, "ee_pose": sm._ee_pose_log, "ee_rpy": sm._ee_rpy_log, "ee_vel": sm._ee_velocity_log,
            "status": sim.solver_status, "sqp_iter": sim.sqp_iter, "qp_iter": sim.qp_iter, "residuals": sim.residuals,
            "cost": sim.cost_history, "solver_time": sim.solver_time, "plant_time": sim.integration_time,
            "errors": analysis_rows(sim)}


def analysis_rows(sim) -> np.ndarray:
    from . import analysis

    return analysis.errors_rows(sim.errors)


def save_results(path: str, names: Sequence[str], configs: Sequence[Dict], records: Sequence[Dict[str, np.ndarray]]) -> str:
    """Write (atomically) one archive for K simulations."""
    K = len(names)
    assert len(configs) == K and len(records) == K
    nsim = np.array([int(np.asarray(r["status"]).shape[0]) for r in records], dtype=np.int64)
    T = int(nsim.max()) if K else 0
    out = {"format_version": np.int64(FORMAT_VERSION), "names": np.array(list(names), dtype=str),
           "configs": np.array([config_json(c) for c in configs], dtype=str),
           "keys": np.array([resume_key(n, c) for n, c in zip(names, configs)], dtype=str), "nsim": nsim}
    for k in LOG_KEYS_T1:
        rows = np.asarray(records[0][k]).shape[0] if K else 0
        a = np.zeros((K, rows, T + 1))
        for i, r in enumerate(records):
            a[i, :, :nsim[i] + 1] = r[k]
        out[k] = a
    for k in LOG_KEYS_T:
        a = np.zeros((K, T), dtype=np.int32 if k in ("status", "sqp_iter", "qp_iter") else np.float64)
        for i, r in enumerate(records):
            a[i, :nsim[i]] = r[k]
        out[k] = a
    a = np.zeros((K, T, 4))
    for i, r in enumerate(records):
        a[i, :nsim[i]] = r["residuals"]
    out["residuals"] = a
    tmp = path + ".tmp.npz"
    np.savez_compressed(tmp, **out)
    os.replace(tmp, path)
    return path


def load_archive(path: str) -> Dict[str, np.ndarray]:
    """Format 2, or format 1 (written before `errors` / `plant_time` were logged): a v1 archive holds everything they
    derive from, so the task errors are recomputed from its pose / velocity logs (analysis.compute_errors) and
    plant_time is zero -- a grid search checkpointed by the older version still resumes."""
    with np.load(path, allow_pickle=False) as f:
        ver = int(f["format_version"])
        if ver not in (1, FORMAT_VERSION):
            raise ValueError(f"{path}: unsupported format version {ver}")
        arch = {k: f[k] for k in f.files}
    if ver == 1:
        from . import analysis
        from .config import resolve_config

        K = len(arch["names"])
        T1 = arch["z"].shape[-1] if K else 1
        err = np.zeros((K, 7, T1))
        for i in range(K):
            c = resolve_config(json.loads(str(arch["configs"][i])))
            n = int(arch["nsim"][i])
            e = analysis.compute_errors(arch["ee_pose"][i][:, :n + 1], arch["ee_vel"][i][:, :n + 1], c["coeffs"], c["t_ee"],
                                        c["px_ref"], c["vy_ref"])
            err[i, :, :n + 1] = analysis.errors_rows(e)
        arch["errors"] = err
        arch["plant_time"] = np.zeros_like(arch["solver_time"])
        arch["format_version"] = np.int64(FORMAT_VERSION)
    return arch


def record_at(arch: Dict[str, np.ndarray], i: int) -> Dict[str, np.ndarray]:
    n = int(arch["nsim"][i])
    rec = {k: arch[k][i][:, :n + 1].copy() for k in LOG_KEYS_T1}
    rec.update({k: arch[k][i][:n].copy() for k in LOG_KEYS_T})
    rec["residuals"] = arch["residuals"][i][:n].copy()
    return rec


def load_results(path: str) -> List[Dict]:
    """Archive -> the list ``run_all`` returns: {'name','simulator','data','analysis','summary'}."""
    from .simulator import Simulator

    arch = load_archive(path)
    out = []
    for i, name in enumerate(arch["names"]):
        sim = Simulator(**json.loads(str(arch["configs"][i])))
        sim.name = str(name)
        sim._attach(record_at(arch, i))
        out.append({"name": sim.name, "simulator": sim, "data": sim.get_data(), "analysis": sim.get_analysis(),
                    "summary": sim.get_summary()})
    return out
