"""ctypes binding of libmpcbatch.so (include/mpcbatch.h) -- the only compute path.

This is the layer that stands where ``acados_template.AcadosOcpSolver`` stands in the
reference (trajectory_optimizer.py:183-186; call sites simulator.py:210-221): a thin
ctypes wrapper around a native solver library.  PyTorch is used as plumbing only: it owns
the device result buffers and the HIP stream, so results can be gathered across GPUs with
``torch.distributed`` (RCCL) without a host round trip.

There is no CPU fallback: if the library or a GPU is missing, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import packing

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmpcbatch.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

RESULT_FIELDS = (
    # name, ctype, per-simulation shape as a function of (Nsim, T1)
    ("z", "f8", lambda S, T: (12, T)), ("u", "f8", lambda S, T: (6, T)), ("ee_pose", "f8", lambda S, T: (12, T)),
    ("ee_rpy", "f8", lambda S, T: (3, T)), ("ee_vel", "f8", lambda S, T: (6, T)),
    ("status", "i4", lambda S, T: (S,)), ("sqp_iter", "i4", lambda S, T: (S,)), ("qp_iter", "i4", lambda S, T: (S,)),
    ("residuals", "f8", lambda S, T: (S, 4)), ("cost", "f8", lambda S, T: (S,)), ("solver_time", "f8", lambda S, T: (S,)),
    ("errors", "f8", lambda S, T: (7, T)), ("plant_time", "f8", lambda S, T: (S,)),
)
ERROR_ROWS = ("e1", "e2", "e3", "e4", "e5", "p_task_z", "p_ee_y")   # rows of `errors` (simulator.py:337-344)
NSUMMARY = 24
# columns of the mpcb_summary record (include/mpcbatch.h)
SUMMARY_COLS = ("rmse_e1", "rmse_e2", "rmse_e3", "rmse_e4", "rmse_e5", "itse_e1", "itse_e2", "itse_e3", "itse_e4", "itse_e5",
                "weighted_rmse", "total_sqp_iterations", "avg_sqp_iterations", "num_failures", "max_kkt_residual",
                "total_solver_time", "avg_mpc_time", "avg_solver_time", "avg_integration_time", "total_computation_time",
                "total_qp_iterations")


class MpcbProblem(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("batch", "N", "Nsim", "solver_type", "max_iter", "qp_iter_max", "fixed_step",
                                       "precision")]


class MpcbResult(C.Structure):
    _fields_ = [("z", _dp), ("u", _dp), ("ee_pose", _dp), ("ee_rpy", _dp), ("ee_vel", _dp), ("status", _ip),
                ("sqp_iter", _ip), ("qp_iter", _ip), ("residuals", _dp), ("cost", _dp), ("solver_time", _dp),
                ("errors", _dp), ("plant_time", _dp)]


class EngineError(RuntimeError):
    pass


_EXPORTS = ("mpcb_version", "mpcb_device_count", "mpcb_create", "mpcb_destroy", "mpcb_last_error",
            "mpcb_workspace_bytes", "mpcb_result_bytes_per_sim", "mpcb_setup", "mpcb_rollout", "mpcb_sync",
            "mpcb_last_kernel_ms", "mpcb_kernel_info", "mpcb_launch_info", "mpcb_engine", "mpcb_engine_for", "mpcb_summary",
            "mpcb_run")


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen libmpcbatch.so and declare the signatures of include/mpcbatch.h."""
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise EngineError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          f"(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch bundles its own libamdhip64.so.7; import it first so that libmpcbatch.so binds to the
    # SAME HIP runtime instance (two runtimes in one process cannot share the device).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name in _EXPORTS:
        if not hasattr(lib, name):
            raise EngineError(f"{path} does not export {name}")
    lib.mpcb_version.restype = C.c_int
    lib.mpcb_device_count.restype = C.c_int
    lib.mpcb_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.mpcb_destroy.argtypes = [C.c_void_p]
    lib.mpcb_destroy.restype = None
    lib.mpcb_last_error.argtypes = [C.c_void_p]
    lib.mpcb_last_error.restype = C.c_char_p
    lib.mpcb_engine_for.argtypes = [C.POINTER(MpcbProblem)]
    lib.mpcb_engine_for.restype = C.c_int
    lib.mpcb_workspace_bytes.argtypes = [C.POINTER(MpcbProblem)]
    lib.mpcb_workspace_bytes.restype = C.c_size_t
    lib.mpcb_result_bytes_per_sim.argtypes = [C.POINTER(MpcbProblem)]
    lib.mpcb_result_bytes_per_sim.restype = C.c_size_t
    lib.mpcb_setup.argtypes = [C.c_void_p, C.POINTER(MpcbProblem), _dp, _dp]
    lib.mpcb_rollout.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(MpcbResult), C.c_void_p]
    lib.mpcb_sync.argtypes = [C.c_void_p]
    lib.mpcb_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.mpcb_kernel_info.argtypes = [C.c_void_p] + [_ip] * 4
    lib.mpcb_run.argtypes = [C.c_void_p, C.POINTER(MpcbProblem), _dp, _dp, C.POINTER(MpcbResult)]
    lib.mpcb_summary.argtypes = [C.c_void_p, C.POINTER(MpcbResult), _dp, C.c_void_p]
    lib.mpcb_launch_info.argtypes = [C.c_void_p, _ip, _ip]
    lib.mpcb_engine.argtypes = [C.c_void_p]
    if hasattr(lib, "mpcb_debug_task_lin"):
        lib.mpcb_debug_task_lin.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp]
    return lib


def make_problem(cfgs: Sequence[Dict]) -> MpcbProblem:
    c0 = cfgs[0]
    ragged = any(c["N"] != c0["N"] for c in cfgs)
    if ragged and c0["solver_type"] != packing.SOLVER_RTI:
        raise ValueError("simulations of different horizons share a launch only with nlp_solver_type='SQP_RTI'")
    key0 = packing.bucket_key(c0, ragged)
    for c in cfgs[1:]:
        if packing.bucket_key(c, ragged) != key0:
            raise ValueError("all simulations of one launch must share Nsim, solver options and robot (and N, except SQP_RTI)")
    return MpcbProblem(len(cfgs), max(c["N"] for c in cfgs), c0["Nsim"], c0["solver_type"], c0["max_iter"], c0["qp_iter_max"],
                       int(c0["fixed_step"]), int(c0.get("precision", 0)))


def engine_for(batch: int, N: int, Nsim: int, solver: str = "SQP_RTI", precision: int = 0, lib: Optional[C.CDLL] = None) -> int:
    """Kernel family a uniform bucket of this shape is sent to (mpcb_engine_for; host logic, no GPU needed):
    0 latency engine, 1 throughput engine."""
    lib = lib or load_library()
    pb = MpcbProblem(batch, N, Nsim, 0 if solver == "SQP" else 1, 100, 50, 0, precision)
    return int(lib.mpcb_engine_for(C.byref(pb)))


class MpcBatchEngine:
    """One handle on one GPU.  ``run`` = Simulator.__init__ + Simulator.run for a whole bucket."""

    def __init__(self, device: int = 0, lib_path: Optional[str] = None):
        self.lib = load_library(lib_path)
        self.device = int(device)
        n = self.lib.mpcb_device_count()
        if n <= 0:
            raise EngineError("no HIP device visible: the MPC engine runs on MI355X only (no CPU fallback)")
        h = C.c_void_p()
        rc = self.lib.mpcb_create(C.byref(h), self.device)
        if rc != 0:
            raise EngineError(f"mpcb_create(device={device}) failed with {rc} ({n} device(s) visible)")
        self._h = h
        self.last_kernel_ms: List[float] = []

    def close(self):
        if getattr(self, "_h", None):
            self.lib.mpcb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.mpcb_last_error(self._h)
            raise EngineError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def kernel_info(self) -> Dict[str, int]:
        v = [C.c_int(0) for _ in range(4)]
        self._check(self.lib.mpcb_kernel_info(self._h, *[C.byref(x) for x in v]), "mpcb_kernel_info")
        return dict(vgprs=v[0].value, sgprs=v[1].value, lds_bytes=v[2].value, scratch_bytes=v[3].value)

    def launch_info(self) -> Dict[str, int]:
        """Launch geometry chosen by setup(): kernel family (0 latency, 1 throughput engine), wavefronts per
        simulation, LDS chunk pool bytes."""
        w, pbytes = C.c_int(0), C.c_int(0)
        self._check(self.lib.mpcb_launch_info(self._h, C.byref(w), C.byref(pbytes)), "mpcb_launch_info")
        return dict(waves_per_sim=w.value, pool_bytes=pbytes.value, engine=int(self.lib.mpcb_engine(self._h)))

    # ------------------------------------------------------------------ device-resident path
    @staticmethod
    def prepare(cfgs: Sequence[Dict], chain):
        """Resolved configs -> what mpcb_setup takes: (problem, parameter records [batch, NPARAM], robot constants [105]).
        This is the construction-time half of Simulator.__init__ (dict -> numbers); nothing touches the device."""
        pb = make_problem(cfgs)
        params = packing.pack_batch(cfgs)
        robot = np.ascontiguousarray(chain.packed(cfgs[0]["t_ee"]), dtype=np.float64)
        assert params.shape == (len(cfgs), packing.NPARAM) and robot.shape == (105,)
        return pb, params, robot

    def setup_packed(self, pb: MpcbProblem, params: np.ndarray, robot: np.ndarray) -> MpcbProblem:
        """mpcb_setup: validate, derive the model constants, upload the records (H2D), size the workspace."""
        self._check(self.lib.mpcb_setup(self._h, C.byref(pb), params.ctypes.data_as(_dp), robot.ctypes.data_as(_dp)),
                    "mpcb_setup")
        self._pb = pb
        return pb

    def setup(self, cfgs: Sequence[Dict], chain) -> MpcbProblem:
        return self.setup_packed(*self.prepare(cfgs, chain))

    def alloc_results(self, pb: MpcbProblem):
        """Device result buffers as torch tensors (dict name -> tensor [batch, ...])."""
        import torch

        dev = torch.device("cuda", self.device)
        S, T = pb.Nsim, pb.Nsim + 1
        out = {}
        for name, ty, shp in RESULT_FIELDS:
            dt = torch.float64 if ty == "f8" else torch.int32
            out[name] = torch.zeros((pb.batch,) + shp(S, T), dtype=dt, device=dev)
        return out

    @staticmethod
    def _result_struct(bufs) -> MpcbResult:
        r = MpcbResult()
        for name, ty, _ in RESULT_FIELDS:
            ptr = bufs[name].data_ptr() if hasattr(bufs[name], "data_ptr") else bufs[name].ctypes.data
            setattr(r, name, C.cast(C.c_void_p(ptr), _dp if ty == "f8" else _ip))
        return r

    def rollout(self, bufs, step0: int, step1: int, stream: Optional[int] = None):
        """Asynchronous launch of closed-loop steps [step0, step1) into device buffers."""
        if stream is None:
            import torch

            stream = torch.cuda.current_stream(self.device).cuda_stream
        r = self._result_struct(bufs)
        self._check(self.lib.mpcb_rollout(self._h, step0, step1, C.byref(r), C.c_void_p(stream)), "mpcb_rollout")

    def sync(self):
        self._check(self.lib.mpcb_sync(self._h), "mpcb_sync")

    def summary(self, bufs, stream: Optional[int] = None):
        """Per-simulation summary record [batch, NSUMMARY] (device tensor) of a finished rollout --
        Simulator.metrics / solver_stats / timings / get_summary for the whole batch in one launch."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        out = torch.empty((self._pb.batch, NSUMMARY), dtype=torch.float64, device=torch.device("cuda", self.device))
        r = self._result_struct(bufs)
        self._check(self.lib.mpcb_summary(self._h, C.byref(r), C.cast(C.c_void_p(out.data_ptr()), _dp), C.c_void_p(stream)),
                    "mpcb_summary")
        return out

    def debug_task_lin(self, cfgs: Sequence[Dict], chain, x: np.ndarray) -> np.ndarray:
        """Diagnostic: the device linearisation at points x[i] = [q; qdot] with the parameters of cfgs[i];
        returns [n, 60] records laid out like a G2 stage record (r 0..4, dg/dq 24..53, dg5/dqdot 54..59)."""
        n = len(cfgs)
        params = packing.pack_batch(cfgs)
        robot = np.ascontiguousarray(chain.packed(cfgs[0]["t_ee"]), dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(n, 12)
        rec = np.zeros((n, 60))
        self._check(self.lib.mpcb_debug_task_lin(self._h, n, params.ctypes.data_as(_dp), robot.ctypes.data_as(_dp),
                                                 x.ctypes.data_as(_dp), rec.ctypes.data_as(_dp)), "mpcb_debug_task_lin")
        return rec

    def kernel_ms(self) -> float:
        ms = C.c_float(0)
        self._check(self.lib.mpcb_last_kernel_ms(self._h, C.byref(ms)), "mpcb_last_kernel_ms")
        return float(ms.value)

    def run_device(self, cfgs: Sequence[Dict], chain, step_chunk: int = 0):
        """setup + all steps; returns (problem, dict of device tensors). Synchronous."""
        pb = self.setup(cfgs, chain)
        bufs = self.alloc_results(pb)
        chunk = step_chunk if step_chunk and step_chunk > 0 else pb.Nsim
        self.last_kernel_ms = []
        for s0 in range(0, pb.Nsim, chunk):
            self.rollout(bufs, s0, min(pb.Nsim, s0 + chunk))
            self.sync()
            self.last_kernel_ms.append(self.kernel_ms())
        return pb, bufs

    def run(self, cfgs: Sequence[Dict], chain, step_chunk: int = 0) -> Dict[str, np.ndarray]:
        """Whole bucket, results as numpy arrays [batch, ...] (D2H copy included)."""
        _, bufs = self.run_device(cfgs, chain, step_chunk)
        return {k: v.cpu().numpy() for k, v in bufs.items()}

    # ------------------------------------------------------------------ host-buffer path
    def run_host_buffers(self, cfgs: Sequence[Dict], chain) -> Dict[str, np.ndarray]:
        """Same through mpcb_run (library-owned device buffers, numpy in/out, no torch)."""
        pb = make_problem(cfgs)
        params = packing.pack_batch(cfgs)
        robot = np.ascontiguousarray(chain.packed(cfgs[0]["t_ee"]), dtype=np.float64)
        S, T = pb.Nsim, pb.Nsim + 1
        out = {name: np.zeros((pb.batch,) + shp(S, T), dtype=np.float64 if ty == "f8" else np.int32)
               for name, ty, shp in RESULT_FIELDS}
        r = self._result_struct(out)
        self._check(self.lib.mpcb_run(self._h, C.byref(pb), params.ctypes.data_as(_dp), robot.ctypes.data_as(_dp),
                                      C.byref(r)), "mpcb_run")
        return out
