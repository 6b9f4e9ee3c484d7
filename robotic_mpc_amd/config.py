"""Configuration handling: the reference's ``BASE_PARAMS`` dict -> packed parameters.

In the reference the config dict is splatted into ``Simulator(**config)``
(simulator.py:18-35, 658-659) and from there into ``MPC(...)`` and the acados option
object (trajectory_optimizer.py:57-70; simulator.py:129-135).  Here the same dict is
validated and flattened into the per-instance parameter record the C-ABI takes; nothing
is code-generated.
"""
from __future__ import annotations

import copy
import warnings
from typing import Any, Dict, Mapping, Optional

import numpy as np

# Positional-without-default arguments of Simulator.__init__ (simulator.py:18-27).
REQUIRED_KEYS = (
    "robot_name", "dt", "simulation_time", "prediction_horizon", "q_0", "qdot_0", "wcv",
    "q_min", "q_max", "qdot_min", "qdot_max", "surface_limits", "surface_origin",
    "surface_orientation_rpy", "w_qddot",
)
# Keyword arguments with defaults (simulator.py:30-35).
OPTIONAL_DEFAULTS = {
    "surface_coeffs": None, "solver_options": None, "w_u": 0.01, "px_ref": 0.40, "vy_ref": 0.05, "scene": True,
}
# Extensions beyond the reference's signature: all optional.
EXTENSION_DEFAULTS = {"translation_ee_t": (0.0, 0.0, 0.1), "urdf_path": None, "ee_frame": None,
                      # plant integrator of simulation_model.Robot (simulation_model.py:13,39-51); the reference's
                      # Simulator hard-codes "RK4" (simulator.py:85)
                      "integration_method": "RK4",
                      # arithmetic of the Riccati factor / solve sweeps: "fp64" (the reference's) or "fp32"
                      # (BASELINE configs[4]; SQP_RTI only)
                      "riccati_precision": "fp64",
                      # bound-inactive fast path of the QP solve (csrc/mpc_ipm.h): a QP whose equality-constrained minimiser is strictly
                      # inside every bound is solved by ONE Riccati factorisation instead of the interior-point loop (same solution to
                      # well below qp_tol).  False: every QP goes through the HPIPM-style loop, as in the reference's solver.solve()
                      # (simulator.py:212)
                      "qp_fast_path": True}
# codes of the parameter record (include/mpcbatch.h [7]); RK4 = 0 keeps default records unchanged
PLANT_INTEGRATORS = {"RK4": 0, "Euler": 1, "RK2": 2, "RK3": 3}

# surface.py:14-17
DEFAULT_SURFACE_COEFFS = {"a": -0.15, "b": 0.15, "c": -0.01, "d": 0.01, "e": 0.01, "f": 0.0}
# trajectory_optimizer.py:44-48
TASK_WEIGHTS = (50.0, 50.0, 50.0, 50.0, 50.0)

SOLVER_SQP, SOLVER_RTI = 0, 1

# acados options set by MPC.__init__ (trajectory_optimizer.py:60-70) + acados defaults.
DEFAULT_SOLVER_OPTIONS = {
    "nlp_solver_type": "SQP",
    "hessian_approx": "GAUSS_NEWTON",
    "integrator_type": "DISCRETE",
    "qp_solver": "PARTIAL_CONDENSING_HPIPM",
    "qp_tol": 1e-8,
    "tol": 1e-6,
    "nlp_solver_max_iter": 100,
    "qp_solver_iter_max": 50,
    "globalization": "MERIT_BACKTRACKING",
    "qp_solver_warm_start": 2,
    "nlp_solver_warm_start_first_qp": True,
    "print_level": 0,
    # acados defaults of options the reference never sets but forwards when given (simulator.py:129-135)
    "levenberg_marquardt": 0.0,
    "nlp_solver_tol_stat": 1e-6, "nlp_solver_tol_eq": 1e-6, "nlp_solver_tol_ineq": 1e-6, "nlp_solver_tol_comp": 1e-6,
}
_NLP_TOLS = ("nlp_solver_tol_stat", "nlp_solver_tol_eq", "nlp_solver_tol_ineq", "nlp_solver_tol_comp")
_QP_TOLS = ("qp_solver_tol_stat", "qp_solver_tol_eq", "qp_solver_tol_ineq", "qp_solver_tol_comp")
# Options that select HOW the QP is solved, not WHAT it is: the GN QP is strictly convex, so its solution
# (to qp_tol) does not depend on condensing or on the Riccati variant.  The engine has one
# QP method (Mehrotra IPM on the sparse OCP-QP, HPIPM semantics); these are accepted without effect.
_ACCEPTED_NOOP = {
    "qp_solver_cond_N", "print_level", "N_horizon", "tf", "qp_solver_cond_ric_alg", "qp_solver_ric_alg",
    "ext_fun_compile_flags", "regularize_method",   # (regularize_method: only NO_REGULARIZE passes the check below)
}
_QP_SOLVERS_EQUIVALENT = ("PARTIAL_CONDENSING_HPIPM", "FULL_CONDENSING_HPIPM")
# Attributes that DO exist on acados' AcadosOcpOptions and change the SQP / interior-point path.  The reference
# forwards any existing attribute by setattr (simulator.py:129-135), so acados would honour them; the engine
# implements exactly one value of each (acados' default, listed here) and REFUSES any other value instead of
# handing back silently different iteration counts and trajectories.  Old and new (globalization_*) spellings.
_FIXED_BEHAVIOUR = {
    "alpha_min": 0.05, "globalization_alpha_min": 0.05,
    "alpha_reduction": 0.7, "globalization_alpha_reduction": 0.7,
    "nlp_solver_step_length": 1.0, "globalization_fixed_step_length": 1.0,
    "line_search_use_sufficient_descent": 0, "globalization_line_search_use_sufficient_descent": 0,
    "eps_sufficient_descent": 1e-4, "globalization_eps_sufficient_descent": 1e-4,
    "full_step_dual": 0, "globalization_full_step_dual": 0,
    "globalization_use_SOC": 0, "globalization_funnel_use_merit_fun_only": False,
    "hpipm_mode": "BALANCE", "qp_solver_mu0": 0.0, "qp_solver_t0_init": 2,
    "reg_epsilon": 1e-4, "reg_max_cond_block": 1e7, "reg_adaptive_eps": False, "reg_min_epsilon": 1e-8,
    "cost_discretization": "EULER", "cost_scaling": None, "fixed_hess": 0,
    "exact_hess_cost": 1, "exact_hess_dyn": 1, "exact_hess_constr": 1,
    "nlp_solver_ext_qp_res": 0, "rti_phase": 0, "as_rti_iter": 1, "as_rti_level": 4, "rti_log_residuals": 0,
    "rti_log_only_available_residuals": 0, "qp_solver_cond_block_size": None, "nlp_qp_tol_strategy": "FIXED_QP_TOL",
    "with_adaptive_levenberg_marquardt": False, "adaptive_levenberg_marquardt_lam": 5.0,
    "adaptive_levenberg_marquardt_mu_min": 1e-16, "adaptive_levenberg_marquardt_mu0": 1e-3,
    "store_iterates": False, "timeout_max_time": 0.0, "qpscaling_scale_objective": "NO_OBJECTIVE_SCALING",
    "qpscaling_scale_constraints": "NO_CONSTRAINT_SCALING", "tau_min": 0.0,
    "sim_method_num_stages": 4, "sim_method_num_steps": 1, "sim_method_newton_iter": 3, "sim_method_jac_reuse": 0,
    "collocation_type": "GAUSS_LEGENDRE", "time_steps": None, "shooting_nodes": None,
}


def _same_option_value(value, default) -> bool:
    if default is None:
        return value is None
    if isinstance(default, str):
        return value == default
    try:
        return float(value) == float(default)
    except (TypeError, ValueError):
        return False

# The common synthetic base of SURVEY.md 8(d) (readme.md:27-43, surface_stats.ipynb cell 1).
BASE_PARAMS: Dict[str, Any] = {
    "robot_name": "ur10",
    "dt": 0.01,
    "simulation_time": 6,
    "prediction_horizon": 100,
    "surface_limits": ((-2, 2), (-2, 2)),
    "surface_origin": np.array([0.0, 0.0, 0.0]),
    "surface_orientation_rpy": np.array([0.0, 0.0, 0.0]),
    "q_0": np.array([np.pi / 4, -np.pi / 3, np.pi / 4, -np.pi / 2, -np.pi / 2, 0.0]),
    "qdot_0": np.array([1.0, 2.0, 1.0, 0.0, 0.0, 0.0]),
    "wcv": np.array([200.0] * 6),
    "q_min": np.array([-2 * np.pi] * 6),
    "q_max": np.array([+2 * np.pi] * 6),
    "qdot_min": np.array([-2.16, -2.16, -np.pi, -3.20, -3.20, -3.20]),
    "qdot_max": np.array([2.16, 2.16, np.pi, 3.20, 3.20, 3.20]),
    "w_qddot": 0.02,
    "w_u": 0.01,
    "px_ref": 0.40,
    "vy_ref": 0.05,
    "solver_options": {"nlp_solver_type": "SQP_RTI"},
    "scene": False,
}


def base_params(**overrides) -> Dict[str, Any]:
    """A deep copy of BASE_PARAMS with overrides applied."""
    cfg = copy.deepcopy(BASE_PARAMS)
    cfg.update(overrides)
    return cfg


def _vec6(cfg: Mapping[str, Any], key: str) -> np.ndarray:
    v = np.asarray(cfg[key], dtype=np.float64).reshape(-1)
    if v.shape != (6,):
        raise ValueError(f"config['{key}'] must have 6 entries, got shape {np.shape(cfg[key])}")
    if not np.all(np.isfinite(v)):
        raise ValueError(f"config['{key}'] must be finite")
    return v


def resolve_solver_options(options: Optional[Mapping[str, Any]]) -> Dict[str, Any]:
    """Mirror of Simulator._apply_solver_options (simulator.py:129-135): known attributes are
    set, unknown ones only warn."""
    out = dict(DEFAULT_SOLVER_OPTIONS)
    if options is None:
        return out
    if not isinstance(options, Mapping):
        raise TypeError("solver_options must be a dict (simulator.py:131 iterates .items())")
    qp_tols = {}
    for key, value in options.items():   # in dict order, like the setattr loop of the reference
        if key == "tol":
            # acados_template's `tol` setter writes all four NLP tolerances
            out["tol"] = value
            for k in _NLP_TOLS:
                out[k] = value
        elif key in _QP_TOLS:
            qp_tols[key] = value
        elif key in out or key in _ACCEPTED_NOOP:
            out[key] = value
        elif key in _FIXED_BEHAVIOUR:
            if not _same_option_value(value, _FIXED_BEHAVIOUR[key]):
                raise ValueError(f"solver option {key}={value!r}: acados would honour it (simulator.py:129-135) and it changes the "
                                 f"SQP / QP iterations; the engine implements {key}={_FIXED_BEHAVIOUR[key]!r} only")
        else:
            warnings.warn(f"Warning: Unknown solver option '{key}'")
    # options that would change results and that the engine does not implement are errors, never ignored
    for key, value in qp_tols.items():
        if float(value) != float(out["qp_tol"]):
            raise ValueError(f"{key}={value} differs from qp_tol={out['qp_tol']}: the engine's interior point takes one "
                             f"tolerance for all four QP residuals (set qp_tol)")
    if out["qp_solver"] not in _QP_SOLVERS_EQUIVALENT:
        raise ValueError(f"qp_solver '{out['qp_solver']}' is not available: the engine solves the OCP-QP with an HPIPM-style "
                         f"interior point ({', '.join(_QP_SOLVERS_EQUIVALENT)} give the same solution to qp_tol)")
    if int(out["qp_solver_warm_start"]) != 2 or not out["nlp_solver_warm_start_first_qp"]:
        raise ValueError("only qp_solver_warm_start=2 with nlp_solver_warm_start_first_qp=True is implemented "
                         "(trajectory_optimizer.py:69-70)")
    if "regularize_method" in options and options["regularize_method"] not in (None, "NO_REGULARIZE"):
        raise ValueError("regularize_method other than NO_REGULARIZE is not implemented")
    for k in _NLP_TOLS + ("qp_tol", "levenberg_marquardt"):
        out[k] = float(out[k])
        if not out[k] >= 0.0 or (k != "levenberg_marquardt" and out[k] == 0.0):
            raise ValueError(f"solver option {k} must be positive")
    if out["nlp_solver_type"] not in ("SQP", "SQP_RTI"):
        raise ValueError(f"nlp_solver_type '{out['nlp_solver_type']}' not supported (SQP, SQP_RTI)")
    if out["hessian_approx"] != "GAUSS_NEWTON":
        raise ValueError("only hessian_approx='GAUSS_NEWTON' is implemented (trajectory_optimizer.py:61)")
    if out["integrator_type"] != "DISCRETE":
        raise ValueError("only integrator_type='DISCRETE' is implemented (trajectory_optimizer.py:64)")
    if out["globalization"] not in ("MERIT_BACKTRACKING", "FIXED_STEP"):
        raise ValueError(f"globalization '{out['globalization']}' not supported")
    return out


def resolve_config(config: Mapping[str, Any]) -> Dict[str, Any]:
    """Validate a Simulator(**config) dict and flatten it to solver parameters.

    Raises TypeError for missing/unknown keys exactly where ``Simulator(**config)`` would.
    """
    missing = [k for k in REQUIRED_KEYS if k not in config]
    if missing:
        raise TypeError(f"Simulator.__init__() missing required arguments: {missing}")
    known = set(REQUIRED_KEYS) | set(OPTIONAL_DEFAULTS) | set(EXTENSION_DEFAULTS)
    unknown = [k for k in config if k not in known]
    if unknown:
        raise TypeError(f"Simulator.__init__() got unexpected keyword arguments: {unknown}")
    cfg = {**OPTIONAL_DEFAULTS, **EXTENSION_DEFAULTS, **config}

    dt = float(cfg["dt"])
    if not dt > 0:
        raise ValueError("dt must be positive")
    Nsim = int(float(cfg["simulation_time"]) / dt)  # simulator.py:41
    N = int(cfg["prediction_horizon"])
    if N < 1:
        raise ValueError("prediction_horizon must be >= 1")
    if Nsim < 1:
        raise ValueError("simulation_time/dt must give at least one step")
    coeffs = dict(DEFAULT_SURFACE_COEFFS)
    if cfg["surface_coeffs"]:
        extra = set(cfg["surface_coeffs"]) - set(coeffs)
        if extra:
            raise KeyError(f"unknown surface coefficient(s) {sorted(extra)}; expected a..f")
        coeffs.update(cfg["surface_coeffs"])  # surface.py:18-19
    so = resolve_solver_options(cfg["solver_options"])
    wcv = _vec6(cfg, "wcv")
    if np.any(wcv <= 0):
        raise ValueError("wcv must be positive (prediction_model.py:94 divides by it)")
    if cfg["integration_method"] not in PLANT_INTEGRATORS:
        raise ValueError(f"Unknown integration method: {cfg['integration_method']}")  # simulation_model.py:51
    if cfg["riccati_precision"] not in ("fp64", "fp32"):
        raise ValueError("riccati_precision must be 'fp64' or 'fp32'")
    if cfg["riccati_precision"] == "fp32" and so["nlp_solver_type"] != "SQP_RTI":
        raise ValueError("riccati_precision='fp32' is implemented for nlp_solver_type='SQP_RTI' only")
    t_ee = np.asarray(cfg["translation_ee_t"], dtype=np.float64).reshape(-1)
    if t_ee.shape != (3,):
        raise ValueError("translation_ee_t must have 3 entries")
    return {
        "robot_name": cfg["robot_name"], "urdf_path": cfg["urdf_path"], "ee_frame": cfg["ee_frame"],
        "N": N, "Nsim": Nsim, "dt": dt,
        "solver_type": SOLVER_RTI if so["nlp_solver_type"] == "SQP_RTI" else SOLVER_SQP,
        "max_iter": int(so["nlp_solver_max_iter"]), "qp_iter_max": int(so["qp_solver_iter_max"]),
        # `tol` of the parameter record is nlp_solver_tol_stat; the other three travel beside it
        "tol": so["nlp_solver_tol_stat"], "tol_eq": so["nlp_solver_tol_eq"], "tol_ineq": so["nlp_solver_tol_ineq"],
        "tol_comp": so["nlp_solver_tol_comp"], "qp_tol": so["qp_tol"], "levenberg_marquardt": so["levenberg_marquardt"],
        "fixed_step": so["globalization"] == "FIXED_STEP",
        "precision": 1 if cfg["riccati_precision"] == "fp32" else 0,
        "qp_fast_path": 1 if cfg["qp_fast_path"] else 0,
        "wcv": wcv, "q0": _vec6(cfg, "q_0"), "qdot0": _vec6(cfg, "qdot_0"),
        "qmin": _vec6(cfg, "q_min"), "qmax": _vec6(cfg, "q_max"),
        "umin": _vec6(cfg, "qdot_min"), "umax": _vec6(cfg, "qdot_max"),
        "w_u": float(cfg["w_u"]), "w_qddot": float(cfg["w_qddot"]),
        "px_ref": float(cfg["px_ref"]), "vy_ref": float(cfg["vy_ref"]),
        "coeffs": np.array([coeffs[k] for k in "abcdef"], dtype=np.float64),
        "coeffs_dict": coeffs,
        "w_task": np.array(TASK_WEIGHTS, dtype=np.float64),
        "t_ee": t_ee,
        "integration_method": cfg["integration_method"], "plant_integrator": PLANT_INTEGRATORS[cfg["integration_method"]],
        # stored-but-unused by the OCP, exactly as in the reference
        "surface_limits": cfg["surface_limits"], "surface_origin": cfg["surface_origin"],
        "surface_orientation_rpy": cfg["surface_orientation_rpy"],
    }
