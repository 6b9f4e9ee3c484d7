"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE -- see mpc_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
The product path (robotic_mpc_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmpc_oracle.so")
_lib = None

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


class Robot(C.Structure):
    _fields_ = [("place", C.c_double * 84), ("axis", C.c_double * 18), ("t_ee", C.c_double * 3)]


class Params(C.Structure):
    _fields_ = [
        ("N", C.c_int), ("Nsim", C.c_int), ("solver_type", C.c_int), ("max_iter", C.c_int),
        ("qp_iter_max", C.c_int), ("dt", C.c_double), ("tol", C.c_double), ("qp_tol", C.c_double),
        ("wcv", C.c_double * 6), ("q0", C.c_double * 6), ("qdot0", C.c_double * 6),
        ("qmin", C.c_double * 6), ("qmax", C.c_double * 6), ("umin", C.c_double * 6), ("umax", C.c_double * 6),
        ("w_u", C.c_double), ("w_qddot", C.c_double), ("px_ref", C.c_double), ("vy_ref", C.c_double),
        ("coeffs", C.c_double * 6), ("w_task", C.c_double * 5), ("integrator", C.c_int),
        ("tol_eq", C.c_double), ("tol_ineq", C.c_double), ("tol_comp", C.c_double), ("lm", C.c_double),
        ("fast_path", C.c_int),
    ]


class Output(C.Structure):
    _fields_ = [
        ("z", dp), ("u", dp), ("ee_pose", dp), ("ee_rpy", dp), ("ee_vel", dp),
        ("status", ip), ("sqp_iter", ip), ("qp_iter", ip),
        ("residuals", dp), ("cost", dp), ("solver_time", dp),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "mpc_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmpc_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_run.restype = C.c_int
        _lib.orc_qp_ipm.restype = C.c_int
        _lib.orc_solver_create.restype = C.c_void_p
        _lib.orc_solver_step.restype = C.c_int
    return _lib


def _ptr(a):
    return a.ctypes.data_as(dp)


def make_robot(chain, t_ee=(0.0, 0.0, 0.1)) -> Robot:
    rb = Robot()
    rb.place[:] = list(np.asarray(chain.place, dtype=np.float64).ravel())
    rb.axis[:] = list(np.asarray(chain.axis, dtype=np.float64).ravel())
    rb.t_ee[:] = list(t_ee)
    return rb


def make_params(cfg: Dict) -> Params:
    """cfg: a flat dict as produced by robotic_mpc_amd.config.resolve_config()."""
    p = Params()
    p.N = int(cfg["N"]); p.Nsim = int(cfg["Nsim"]); p.solver_type = int(cfg["solver_type"])
    p.max_iter = int(cfg["max_iter"]); p.qp_iter_max = int(cfg["qp_iter_max"])
    p.dt = float(cfg["dt"]); p.tol = float(cfg["tol"]); p.qp_tol = float(cfg["qp_tol"])
    for k in ("wcv", "q0", "qdot0", "qmin", "qmax", "umin", "umax", "coeffs", "w_task"):
        getattr(p, k)[:] = [float(v) for v in cfg[k]]
    p.w_u = float(cfg["w_u"]); p.w_qddot = float(cfg["w_qddot"])
    p.px_ref = float(cfg["px_ref"]); p.vy_ref = float(cfg["vy_ref"])
    p.integrator = int(cfg.get("plant_integrator", 0))
    p.tol_eq = float(cfg.get("tol_eq", 0.0)); p.tol_ineq = float(cfg.get("tol_ineq", 0.0)); p.tol_comp = float(cfg.get("tol_comp", 0.0))
    p.lm = float(cfg.get("levenberg_marquardt", 0.0))
    p.fast_path = int(cfg.get("qp_fast_path", 1))
    return p


def fk(rb: Robot, q) -> np.ndarray:
    out = np.zeros(12)
    lib().orc_fk(C.byref(rb), _ptr(np.ascontiguousarray(q, dtype=np.float64)), _ptr(out))
    return out


def jacobian_world(rb: Robot, q) -> np.ndarray:
    out = np.zeros((6, 6))
    lib().orc_jacobian_world(C.byref(rb), _ptr(np.ascontiguousarray(q, dtype=np.float64)), _ptr(out))
    return out


def task_output(rb: Robot, q, qd) -> np.ndarray:
    out = np.zeros(15)
    lib().orc_task_output(C.byref(rb), _ptr(np.ascontiguousarray(q, dtype=np.float64)),
                          _ptr(np.ascontiguousarray(qd, dtype=np.float64)), _ptr(out))
    return out


def task_g(rb: Robot, coeffs, q, qd):
    g = np.zeros(5)
    G = np.zeros((5, 12))
    lib().orc_task_g(C.byref(rb), _ptr(np.ascontiguousarray(coeffs, dtype=np.float64)),
                     _ptr(np.ascontiguousarray(q, dtype=np.float64)),
                     _ptr(np.ascontiguousarray(qd, dtype=np.float64)), _ptr(g), _ptr(G))
    return g, G


def stage_residual(rb: Robot, p: Params, x, u):
    r = np.zeros(17)
    Jr = np.zeros((17, 18))
    lib().orc_stage_residual(C.byref(rb), C.byref(p), _ptr(np.ascontiguousarray(x, dtype=np.float64)),
                             _ptr(np.ascontiguousarray(u, dtype=np.float64)), _ptr(r), _ptr(Jr))
    return r, Jr


def plant_step(integrator: int, wcv, dt, z, u) -> np.ndarray:
    """simulation_model.py:93-117: 0 RK4, 1 Euler, 2 RK2, 3 RK3."""
    out = np.zeros(12)
    lib().orc_plant_step(C.c_int(integrator), _ptr(np.ascontiguousarray(wcv, dtype=np.float64)), C.c_double(dt),
                         _ptr(np.ascontiguousarray(z, dtype=np.float64)), _ptr(np.ascontiguousarray(u, dtype=np.float64)),
                         _ptr(out))
    return out


def rk4(wcv, dt, z, u) -> np.ndarray:
    out = np.zeros(12)
    lib().orc_rk4(_ptr(np.ascontiguousarray(wcv, dtype=np.float64)), C.c_double(dt),
                  _ptr(np.ascontiguousarray(z, dtype=np.float64)),
                  _ptr(np.ascontiguousarray(u, dtype=np.float64)), _ptr(out))
    return out


def lti(wcv, Ts):
    a = [np.zeros(6) for _ in range(4)]
    lib().orc_lti(_ptr(np.ascontiguousarray(wcv, dtype=np.float64)), C.c_double(Ts), *[_ptr(v) for v in a])
    return a  # a12, a22, b1, b2


def qp_ipm(H, g, b, A, B, lb, ub, dx0, tol=1e-8, iter_max=50, warm=None):
    """Solve one OCP-QP; returns dict(w, pi, lam, t, status, iters, res)."""
    N = H.shape[0] - 1
    H = np.ascontiguousarray(H, dtype=np.float64); g = np.ascontiguousarray(g, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64); A = np.ascontiguousarray(A, dtype=np.float64)
    B = np.ascontiguousarray(B, dtype=np.float64); lb = np.ascontiguousarray(lb, dtype=np.float64)
    ub = np.ascontiguousarray(ub, dtype=np.float64); dx0 = np.ascontiguousarray(dx0, dtype=np.float64)
    if warm is None:
        w = np.zeros((N + 1, 18)); pi = np.zeros((max(N, 1), 12)); lam = np.zeros((N + 1, 24)); t = np.zeros((N + 1, 24))
    else:
        w, pi, lam, t = (np.ascontiguousarray(v, dtype=np.float64).copy() for v in warm)
    iters = C.c_int(0)
    res = np.zeros(4)
    st = lib().orc_qp_ipm(C.c_int(N), _ptr(H), _ptr(g), _ptr(b), _ptr(A), _ptr(B), _ptr(lb), _ptr(ub), _ptr(dx0),
                          _ptr(w), _ptr(pi), _ptr(lam), _ptr(t), C.c_double(tol), C.c_int(iter_max),
                          C.byref(iters), _ptr(res))
    return dict(w=w, pi=pi, lam=lam, t=t, status=st, iters=iters.value, res=res)


def qp_fast(H, g, b, A, B, lb, ub, dx0, warm=None):
    """The bound-inactive fast path on one OCP-QP; returns dict(accepted, w, pi, lam, t) -- the warm start unchanged when rejected."""
    N = H.shape[0] - 1
    H = np.ascontiguousarray(H, dtype=np.float64); g = np.ascontiguousarray(g, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64); A = np.ascontiguousarray(A, dtype=np.float64)
    B = np.ascontiguousarray(B, dtype=np.float64); lb = np.ascontiguousarray(lb, dtype=np.float64)
    ub = np.ascontiguousarray(ub, dtype=np.float64); dx0 = np.ascontiguousarray(dx0, dtype=np.float64)
    if warm is None:
        w = np.zeros((N + 1, 18)); pi = np.zeros((max(N, 1), 12)); lam = np.zeros((N + 1, 24)); t = np.zeros((N + 1, 24))
    else:
        w, pi, lam, t = (np.ascontiguousarray(v, dtype=np.float64).copy() for v in warm)
    lib().orc_qp_fast.restype = C.c_int
    ok = lib().orc_qp_fast(C.c_int(N), _ptr(H), _ptr(g), _ptr(b), _ptr(A), _ptr(B), _ptr(lb), _ptr(ub), _ptr(dx0),
                           _ptr(w), _ptr(pi), _ptr(lam), _ptr(t))
    return dict(accepted=int(ok), w=w, pi=pi, lam=lam, t=t)


def run(rb: Robot, p: Params) -> Dict[str, np.ndarray]:
    """Closed loop for one instance; arrays shaped like the reference's logs."""
    T1 = p.Nsim + 1
    o = dict(
        z=np.zeros((12, T1)), u=np.zeros((6, T1)), ee_pose=np.zeros((12, T1)), ee_rpy=np.zeros((3, T1)),
        ee_vel=np.zeros((6, T1)), status=np.zeros(p.Nsim, dtype=np.int32), sqp_iter=np.zeros(p.Nsim, dtype=np.int32),
        qp_iter=np.zeros(p.Nsim, dtype=np.int32), residuals=np.zeros((p.Nsim, 4)), cost=np.zeros(p.Nsim),
        solver_time=np.zeros(p.Nsim),
    )
    out = Output()
    for k, v in o.items():
        setattr(out, k, v.ctypes.data_as(ip if v.dtype == np.int32 else dp))
    rc = lib().orc_run(C.byref(rb), C.byref(p), C.byref(out))
    if rc != 0:
        raise RuntimeError(f"orc_run failed: {rc}")
    return o


class Solver:
    """Step-level handle (mirrors AcadosOcpSolver use at simulator.py:210-221)."""

    def __init__(self, rb: Robot, p: Params):
        self._rb, self._p = rb, p
        self._h = C.c_void_p(lib().orc_solver_create(C.byref(rb), C.byref(p)))
        self.N = p.N

    def step(self, xhat):
        u0 = np.zeros(6); res = np.zeros(4)
        sqp = C.c_int(0); qp = C.c_int(0); cost = C.c_double(0)
        st = lib().orc_solver_step(self._h, _ptr(np.ascontiguousarray(xhat, dtype=np.float64)), _ptr(u0),
                                   C.byref(sqp), C.byref(qp), _ptr(res), C.byref(cost))
        return dict(status=st, u0=u0, sqp_iter=sqp.value, qp_iter=qp.value, res=res, cost=cost.value)

    def iterate(self):
        x = np.zeros((self.N + 1, 12)); u = np.zeros((self.N, 6)); pi = np.zeros((self.N, 12))
        lib().orc_solver_get_iterate(self._h, _ptr(x), _ptr(u), _ptr(pi))
        return x, u, pi

    def __del__(self):
        try:
            lib().orc_solver_destroy(self._h)
        except Exception:
            pass
