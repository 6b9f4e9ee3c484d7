"""``Simulator`` / ``SimulationManager`` with the reference's API, backed by the batched
GPU engine.

Mirrors simulator.py of lynet55/robotic-mpc:
  * ``Simulator(**config)``       simulator.py:15-547  (constructor kwargs :18-35, ``run`` :199,
    cached ``errors/metrics/solver_stats/timings`` :265-448, ``get_data/get_analysis/
    get_summary`` :450-547)
  * ``SimulationManager``         simulator.py:549-716 (``sweep`` :562, ``grid_search`` :593,
    ``run_all`` :641, ``clear`` :714) plus the README names ``sweep_parameter`` / ``add_manual``
    (readme.md:66,69).

What changes underneath: the reference constructs, code-generates, compiles and runs one
acados solver per simulation, sequentially (simulator.py:654-674).  Here ``run_all`` packs all
queued configs into parameter records, groups them into buckets that share
(N, Nsim, solver options, robot) and runs every bucket as ONE launch of the HIP engine, one
wavefront per simulation; with ``torch.distributed`` initialised the buckets are sharded over
the ranks (one per GPU) and gathered over RCCL at the end.

Deliberate deviations: no real-time pacing sleeps (C.10), no MeshCat
scene (``scene`` is accepted and ignored), ``grid_search`` deep-copies nested dicts instead
of aliasing them (C.7).
"""
from __future__ import annotations

import copy
import json
import os
import time
from functools import cached_property
from itertools import product
from types import SimpleNamespace
from typing import Any, Callable, Dict, List, Optional, Sequence

import numpy as np

from . import analysis, config as cfgmod, packing, robots


def chain_for(cfg: Dict) -> robots.KinematicChain:
    if cfg.get("urdf_path"):
        frame = cfg.get("ee_frame") or robots._EE_FRAME.get(cfg["robot_name"])
        if frame is None:
            raise ValueError("ee_frame is required with urdf_path for robots other than ur5/ur10")
        return robots.chain_from_urdf(cfg["urdf_path"], frame, cfg["robot_name"])
    return robots.builtin_chain(cfg["robot_name"])


class _PlantLog:
    """Read-only stand-in for simulation_model.Robot's logs and accessors
    (simulation_model.py:25-29, 130-166)."""

    def __init__(self, z, u, ee_pose, ee_rpy, ee_vel, dt):
        self.z, self.u = z, u
        self._ee_pose_log, self._ee_rpy_log, self._ee_velocity_log = ee_pose, ee_rpy, ee_vel
        self._dt, self._N = dt, z.shape[1] - 1

    dt = property(lambda self: self._dt)
    N = property(lambda self: self._N)

    def get_state(self): return self.z[:, -1]
    def get_control_input(self): return self.u[:, -1]
    def ee_position(self, t): return self._ee_pose_log[:3, t]
    def ee_orientation_euler(self, t): return self._ee_rpy_log[:, t]
    def ee_orientation_rot(self, t): return self._ee_pose_log[3:12, t]
    def ee_velocity(self, t): return self._ee_velocity_log[:, t]
    def joint_angles(self, t): return self.z[:6, t]
    def joint_velocities(self, t): return self.z[6:, t]
    def state(self, t): return self.z[:, t]
    def ee_pose(self, t): return self._ee_pose_log[:, t]
    def simulate_output(self): return self._ee_pose_log, self._ee_velocity_log


class Simulator:
    """One closed-loop simulation with the reference's constructor and accessors.

    ``Simulator(**config).run()`` runs a batch of one on the GPU; ``SimulationManager.run_all``
    fills many of these from one batched launch via ``_attach``.
    """

    def __init__(self, robot_name, dt, simulation_time, prediction_horizon, q_0, qdot_0, wcv, q_min, q_max,
                 qdot_min, qdot_max, surface_limits, surface_origin, surface_orientation_rpy, w_qddot,
                 surface_coeffs=None, solver_options=None, w_u=0.01, px_ref=0.40, vy_ref=0.05, scene=True, **extensions):
        self.config = dict(robot_name=robot_name, dt=dt, simulation_time=simulation_time,
                           prediction_horizon=prediction_horizon, q_0=q_0, qdot_0=qdot_0, wcv=wcv, q_min=q_min,
                           q_max=q_max, qdot_min=qdot_min, qdot_max=qdot_max, surface_limits=surface_limits,
                           surface_origin=surface_origin, surface_orientation_rpy=surface_orientation_rpy,
                           w_qddot=w_qddot, surface_coeffs=surface_coeffs, solver_options=solver_options, w_u=w_u,
                           px_ref=px_ref, vy_ref=vy_ref, scene=scene, **extensions)
        self.resolved = cfgmod.resolve_config(self.config)
        r = self.resolved
        # attributes of the reference object (simulator.py:38-65)
        self.name = robot_name
        self.dt = r["dt"]
        self.simulation_time = simulation_time
        self.Nsim = r["Nsim"]
        self.prediction_horizon = r["N"]
        self.q_0, self.qdot_0, self.wcv = r["q0"], r["qdot0"], r["wcv"]
        self.q_min, self.q_max, self.qdot_min, self.qdot_max = r["qmin"], r["qmax"], r["umin"], r["umax"]
        self.w_u, self.w_qddot = r["w_u"], r["w_qddot"]
        self.initial_state = np.hstack((self.q_0, self.qdot_0))
        self.px_ref, self.vy_ref = r["px_ref"], r["vy_ref"]
        self.translation = list(r["t_ee"])
        self.surface = SimpleNamespace(coeffs=dict(r["coeffs_dict"]), limits=surface_limits, position=surface_origin,
                                       orientation_rpy=surface_orientation_rpy)
        w = analysis.TASK_WEIGHT
        self.mpc = SimpleNamespace(px_ref=self.px_ref, vy_ref=self.vy_ref, w_u=self.w_u, w_qddot=self.w_qddot,
                                   N_horizon=self.prediction_horizon, Tf=self.dt * self.prediction_horizon,
                                   w_origin_task=w, w_normal_alignment_task=w, w_x_alignment_task=w,
                                   w_fixed_x_task=w, w_fixed_vy_task=w)
        self.scene = None  # MeshCat is out of scope (simulator.py:120-124 scene=False path)
        self._errors_rows = None
        self._summary_row = None
        self.mpc_time = np.zeros(self.Nsim)
        self.integration_time = np.zeros(self.Nsim)
        self.sqp_iter = np.zeros(self.Nsim, dtype=int)
        self.qp_iter = np.zeros(self.Nsim, dtype=int)
        self.solver_status = np.zeros(self.Nsim, dtype=int)
        self.residuals = np.zeros((self.Nsim, 4))
        self.solver_time = np.zeros(self.Nsim)
        self.cost_history = np.zeros(self.Nsim)
        self.simulation_model: Optional[_PlantLog] = None
        self._data_computed = False

    # ------------------------------------------------------------------ running
    def run(self, engine=None):
        """Run this single simulation on the GPU (simulator.py:199-241)."""
        from .engine import MpcBatchEngine

        eng = engine or MpcBatchEngine(0)
        out = eng.run([self.resolved], chain_for(self.resolved))
        self._attach({k: v[0] for k, v in out.items()})
        return self

    def _attach(self, rec: Dict[str, np.ndarray]):
        """Install one simulation's logs (arrays shaped like the reference's; views into the batch arrays of a
        bucket are fine).  ``rec`` may carry ``errors`` [7, T1] (logged on the device with every column) and
        ``summary`` [24] (mpcb_summary); what is missing is computed on demand from the logs."""
        self._invalidate_cache()
        if "z" not in rec:
            # summary-only result (run_all(results="summary")): the 24-scalar record of mpcb_summary and nothing else left the
            # device; get_summary() / metrics work, the logs and everything derived from them do not exist on the host
            self._summary_row = rec["summary"]
            self._summary_only = True
            return
        self.simulation_model = _PlantLog(rec["z"], rec["u"], rec["ee_pose"], rec["ee_rpy"], rec["ee_vel"], self.dt)
        self.sqp_iter = rec["sqp_iter"]
        self.qp_iter = rec["qp_iter"]
        self.solver_status = rec["status"]
        self.residuals = rec["residuals"]
        self.solver_time = rec["solver_time"]
        self.cost_history = rec["cost"]
        # simulator.py:214,226 time the Python calls around solve() and update(); here both run on the device:
        # mpc_time is the device time of the step's solve (there is no call overhead to add), integration_time
        # the device time of the plant step with its FK / J qdot / error logging
        self.mpc_time = rec["solver_time"]
        self.integration_time = rec["plant_time"] if "plant_time" in rec else np.zeros(self.Nsim)
        self._errors_rows = rec.get("errors")
        self._summary_row = rec.get("summary")
        self._data_computed = True

    def _invalidate_cache(self):
        self._data_computed = False
        for attr in ("errors", "metrics", "solver_stats", "timings"):
            self.__dict__.pop(attr, None)

    def _need_run(self, what):
        if getattr(self, "_summary_only", False):
            if what in ("summary", "metrics"):
                return
            raise RuntimeError(f"{what} needs the logs: this simulation was run with results='summary' (only the summary left the GPU)")
        if not self._data_computed:
            raise RuntimeError(f"Must call run() before accessing {what}")

    # ------------------------------------------------------------------ analysis (cached)
    @cached_property
    def errors(self):
        self._need_run("errors")
        if getattr(self, "_errors_rows", None) is not None:
            return analysis.errors_dict(self._errors_rows)
        sm = self.simulation_model
        return analysis.compute_errors(sm._ee_pose_log, sm._ee_velocity_log, self.resolved["coeffs"], self.translation,
                                       self.px_ref, self.vy_ref)

    @cached_property
    def metrics(self):
        self._need_run("metrics")
        r = getattr(self, "_summary_row", None)
        if r is not None:   # reduced with the batch (mpcb_summary / analysis.batch_summary)
            keys = ("e1", "e2", "e3", "e4", "e5")
            return {"weighted_rmse": float(r[10]), "rmse": {k: float(r[i]) for i, k in enumerate(keys)},
                    "itse": {k: float(r[5 + i]) for i, k in enumerate(keys)}}
        return analysis.compute_metrics(self.errors, self.dt)

    @cached_property
    def solver_stats(self):
        self._need_run("solver_stats")
        return analysis.compute_solver_stats(self.sqp_iter, self.solver_status, self.residuals, self.cost_history,
                                             self.solver_time)

    @cached_property
    def timings(self):
        self._need_run("timings")
        return analysis.compute_timings(self.mpc_time, self.integration_time, self.solver_time)

    def get_data(self):
        self._need_run("data")
        sm = self.simulation_model
        return analysis.compute_data(sm.z, sm.u, sm._ee_pose_log, self.wcv, self.dt, self.mpc.px_ref, self.mpc.vy_ref,
                                     self.prediction_horizon)

    def get_analysis(self):
        self._need_run("analysis")
        return {**self.errors, **self.metrics, **self.solver_stats, **self.timings}

    def get_summary(self):
        """simulator.py:509-547."""
        self._need_run("summary")
        r = getattr(self, "_summary_row", None)
        if r is not None:
            from .engine import SUMMARY_COLS

            out = {k: float(r[i]) for i, k in enumerate(SUMMARY_COLS[:20])}
            for k in ("total_sqp_iterations", "num_failures"):
                out[k] = int(round(out[k]))
            return out
        m, s, t = self.metrics, self.solver_stats, self.timings
        return {
            "rmse_e1": m["rmse"]["e1"], "rmse_e2": m["rmse"]["e2"], "rmse_e3": m["rmse"]["e3"],
            "rmse_e4": m["rmse"]["e4"], "rmse_e5": m["rmse"]["e5"],
            "itse_e1": m["itse"]["e1"], "itse_e2": m["itse"]["e2"], "itse_e3": m["itse"]["e3"],
            "itse_e4": m["itse"]["e4"], "itse_e5": m["itse"]["e5"],
            "weighted_rmse": m["weighted_rmse"],
            "total_sqp_iterations": s["total_sqp_iterations"], "avg_sqp_iterations": s["avg_sqp_iterations"],
            "num_failures": s["num_failures"], "max_kkt_residual": s["max_kkt_residual"],
            "total_solver_time": s["total_solver_time"],
            "avg_mpc_time": t["avg_mpc_time"], "avg_solver_time": t["avg_solver_time"],
            "avg_integration_time": t["avg_integration_time"], "total_computation_time": t["computational_time_sim"],
        }


class _ResultItem(dict):
    """One entry of run_all's list: {'name','simulator','data','analysis','summary'} (simulator.py:668-674).
    'data' and 'analysis' are dicts of arrays derived from the logs; they are built on first access, so a grid
    search of thousands of simulations hands its results back without a per-simulation numpy pass."""

    _LAZY = {"data": "get_data", "analysis": "get_analysis", "summary": "get_summary"}

    def __init__(self, name, sim):
        super().__init__(name=name, simulator=sim)
        # a summary-only result (run_all(results="summary")) has no logs to derive 'data' / 'analysis' from
        self._lazy = {"summary": "get_summary"} if getattr(sim, "_summary_only", False) else self._LAZY

    def __missing__(self, key):
        if key in self._lazy:
            v = getattr(dict.__getitem__(self, "simulator"), self._lazy[key])()
            dict.__setitem__(self, key, v)
            return v
        if key in self._LAZY:
            raise RuntimeError(f"'{key}' needs the logs: this run kept results='summary' (only the summaries left the GPU)")
        raise KeyError(key)

    def _fill(self):
        for k in self._lazy:
            self[k]

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def __contains__(self, key):
        return key in self._lazy or dict.__contains__(self, key)

    def keys(self): self._fill(); return dict.keys(self)
    def items(self): self._fill(); return dict.items(self)
    def values(self): self._fill(); return dict.values(self)
    def __iter__(self): self._fill(); return dict.__iter__(self)
    def __len__(self): return 2 + len(self._lazy)


# a runner maps (list of resolved configs of one bucket, chain) -> dict of arrays [batch, ...]
Runner = Callable[[Sequence[Dict], robots.KinematicChain], Dict[str, np.ndarray]]


class _EngineRunner:
    """Default runner: the HIP engine.  Callable (one bucket, synchronous) and, for queues with several
    buckets, submit()/collect(): every bucket in flight has its own engine handle and HIP stream, so the
    launches of small buckets (fewer simulations than CUs) overlap on the GPU.  collect() returns DEVICE
    tensors (the logs and the per-simulation summary reduced on the device by mpcb_summary): the caller
    gathers them across ranks (RCCL) or copies them to the host, once."""

    max_in_flight = 8   # buckets holding an engine + workspace + result buffers at the same time
    supports_ragged = True   # SQP_RTI buckets that differ only in the horizon may arrive merged (throughput engine)

    def __init__(self, device: int):
        self.device = device
        self._idle: list = []

    def _engine(self):
        from .engine import MpcBatchEngine

        return self._idle.pop() if self._idle else MpcBatchEngine(self.device)

    def __call__(self, cfgs, chain):
        return self.collect(self.submit(cfgs, chain))

    def submit(self, cfgs, chain):
        import torch

        eng = self._engine()
        stream = torch.cuda.Stream(device=self.device)
        pb = eng.setup(cfgs, chain)
        with torch.cuda.stream(stream):
            bufs = eng.alloc_results(pb)
            eng.rollout(bufs, 0, pb.Nsim, stream=stream.cuda_stream)
            summary = eng.summary(bufs, stream=stream.cuda_stream)
        return eng, stream, bufs, summary

    def collect(self, ticket):
        eng, stream, bufs, summary = ticket
        try:
            stream.synchronize()
            eng.sync()                  # (also reports a failed work-queue hand-off of the throughput engine)
            ms = eng.kernel_ms()
        except Exception:
            eng.close()                 # a handle whose launch failed is not recycled
            raise
        self._idle.append(eng)
        out = dict(bufs)
        out["summary"] = summary
        out["_kernel_ms"] = ms
        return out


class SimulationManager:
    """Parameter sweeps and grid searches, run as batched GPU launches."""

    def __init__(self, base_config, runner: Optional[Runner] = None, device: Optional[int] = None):
        self.base_config = base_config.copy()
        self.simulations: List[Dict[str, Any]] = []
        self._runner = runner
        self._device = device
        self.last_run_info: Dict[str, Any] = {}

    # ------------------------------------------------------------------ queueing
    def sweep(self, param_path, values, name_template=None):
        """simulator.py:562-591."""
        for val in values:
            config = self.base_config.copy()
            if "." in param_path:
                parts = param_path.split(".")
                if parts[0] == "surface_coeffs":
                    config["surface_coeffs"] = dict(config.get("surface_coeffs") or {})
                    config["surface_coeffs"][parts[1]] = val
                else:
                    config[parts[0]] = dict(config.get(parts[0]) or {})
                    config[parts[0]][parts[1]] = val
            else:
                config[param_path] = val
            name = name_template.format(val) if name_template else f"{param_path}={val}"
            self.simulations.append({"name": name, "config": config})

    def sweep_parameter(self, name, values, name_template=None):
        """README alias of ``sweep`` (readme.md:66)."""
        return self.sweep(name, values, name_template)

    def add_manual(self, name, params=None, **kw):
        """README ``manager.add_manual(name=..., params={...})`` (readme.md:69): one explicit run
        with ``params`` overriding the base config (dotted keys allowed)."""
        config = copy.deepcopy(self.base_config)
        for k, v in {**(params or {}), **kw}.items():
            if "." in k:
                a, b = k.split(".", 1)
                config[a] = dict(config.get(a) or {})
                config[a][b] = v
            else:
                config[k] = v
        self.simulations.append({"name": name, "config": config})

    def grid_search(self, param_grid, surface_coeff_sets=None, name_template=None):
        """simulator.py:593-639 (nested dicts are copied, not aliased -- SURVEY C.7)."""
        param_paths = list(param_grid.keys())
        value_lists = [param_grid[p] for p in param_paths]
        if surface_coeff_sets is None:
            surface_coeff_sets = [None]
        for coeff_idx, coeff_set in enumerate(surface_coeff_sets):
            for values in product(*value_lists):
                config = self.base_config.copy()
                for param_path, val in zip(param_paths, values):
                    if "." in param_path:
                        parts = param_path.split(".")
                        config[parts[0]] = dict(config.get(parts[0]) or {})
                        config[parts[0]][parts[1]] = val
                    else:
                        config[param_path] = val
                if coeff_set is not None:
                    config["surface_coeffs"] = coeff_set.copy()
                if name_template:
                    try:
                        name = name_template(dict(zip(param_paths, values)), coeff_idx)
                    except TypeError:
                        name = name_template(dict(zip(param_paths, values)))
                else:
                    param_str = "_".join(f"{p.split('.')[-1]}={v}" for p, v in zip(param_paths, values))
                    name = f"coeffs{coeff_idx}_{param_str}" if coeff_set is not None else param_str
                self.simulations.append({"name": name, "config": config})

    def clear(self):
        self.simulations = []

    # ------------------------------------------------------------------ running
    def _default_runner(self) -> Runner:
        from .engine import MpcBatchEngine

        dev = self._device
        if dev is None:
            from .distributed import local_device

            dev = local_device()
        return _EngineRunner(dev)

    def run_all(self, return_results=True, distributed: Optional[bool] = None, checkpoint: Optional[str] = None,
                results: str = "full"):
        """Run every queued simulation (simulator.py:641-676).

        Returns the reference's list of ``{'name','simulator','data','analysis','summary'}``
        dicts, in queue order.  Under ``torch.distributed`` (world_size > 1) every rank must
        call this with the same queue; the full list is returned on rank 0 and ``[]`` elsewhere.

        ``results="summary"`` (extension): only the 21-scalar summary of every simulation
        (``get_summary()``, reduced on the device) is gathered and copied to the host -- 192 B per simulation
        instead of ~262 KB of logs; the items then carry 'name', 'simulator' and 'summary', and asking one for
        'data' / 'analysis' raises.  ``return_results=False`` without a ``checkpoint`` moves the summaries only
        as well (there is nobody to hand the logs to); they stay available as ``self.last_summaries``.

        ``checkpoint`` (extension): path of a results archive (results_io).
        Simulations already in it -- same name and same config -- are not run again; the archive
        is rewritten with the union afterwards, so an interrupted grid search resumes.
        """
        from . import distributed as dmod
        from . import results_io

        t_start = time.time()
        sims = [Simulator(**spec["config"]) for spec in self.simulations]
        for sim, spec in zip(sims, self.simulations):
            sim.name = spec["name"]  # simulator.py:660
        resolved = [s.resolved for s in sims]
        runner = self._runner or self._default_runner()
        use_dist = dmod.is_distributed() if distributed is None else distributed
        done: Dict[int, Dict[str, np.ndarray]] = {}
        arch = None
        if checkpoint and os.path.exists(checkpoint):
            arch = results_io.load_archive(checkpoint)
            have = {str(k): i for i, k in enumerate(arch["keys"])}
            for i, (sim, spec) in enumerate(zip(sims, self.simulations)):
                j = have.get(results_io.resume_key(sim.name, spec["config"]))
                if j is not None:
                    done[i] = results_io.record_at(arch, j)
        todo = [i for i in range(len(sims)) if i not in done]
        info: Dict[str, Any] = {}
        t_run = time.time()
        if results not in ("full", "summary"):
            raise ValueError("results must be 'full' or 'summary'")
        if results == "summary" and checkpoint:
            raise ValueError("a checkpoint archive stores the logs: use results='full' with checkpoint")
        summary_only = results == "summary" or (not return_results and not checkpoint)
        new = dmod.run_partitioned([resolved[i] for i in todo], runner, chain_for, use_dist, info,
                                   only=("summary",) if summary_only else None) if todo else []
        self.last_run_info = {"n_sims": len(sims), "n_resumed": len(done), "setup_s": t_run - t_start,
                              "run_s": time.time() - t_run, "kernel_ms": info.get("kernel_ms", 0.0),
                              "d2h_s": info.get("d2h_s", 0.0),
                              "buckets": len(dmod.group_buckets([resolved[i] for i in todo], dmod.world_size() if use_dist else 1,
                                                                getattr(runner, "supports_ragged", False))),
                              "world_size": dmod.world_size() if use_dist else 1}
        if new is None or (use_dist and dmod.rank() != 0):  # non-root rank
            return [] if return_results else None
        records = [None] * len(sims)
        for i, rec in done.items():
            records[i] = rec
        for i, rec in zip(todo, new):
            records[i] = rec
        if checkpoint and todo:
            names = [s.name for s in sims]
            configs = [spec["config"] for spec in self.simulations]
            recs = list(records)
            if arch is not None:   # keep what the archive holds beyond this queue: a true union
                mine = {results_io.resume_key(n, c) for n, c in zip(names, configs)}
                for j, k in enumerate(arch["keys"]):
                    if str(k) not in mine:
                        names.append(str(arch["names"][j])); configs.append(json.loads(str(arch["configs"][j])))
                        recs.append(results_io.record_at(arch, j))
            results_io.save_results(checkpoint, names, configs, recs)
        out = []
        for sim, rec in zip(sims, records):
            sim._attach(rec)
            if return_results:
                out.append(_ResultItem(sim.name, sim))
        self.last_summaries = [{"name": sim.name, **sim.get_summary()} for sim in sims]
        self.last_run_info["wall_s"] = time.time() - t_start
        self.last_run_info["results"] = "summary" if summary_only else "full"
        return out if return_results else None
