"""Loader for the TEST-ONLY host emulation of the device engine (see emu_harness.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPC_EMU_DEFINES="-DMPCB_REG_MODE=2 ..." builds (and loads) a variant of the engine next to the default one
_DEFINES = os.environ.get("MPC_EMU_DEFINES", "").split()
_LIB = os.path.join(_HERE, "libmpc_emu%s.so" % ("_" + "".join(c if c.isalnum() else "_" for c in "".join(_DEFINES)) if _DEFINES else ""))
_CSRC = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "robotic_mpc_amd", "csrc")


class Problem(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("batch", "N", "Nsim", "solver_type", "max_iter", "qp_iter_max", "fixed_step", "precision")]


def build(force=False):
    srcs = [os.path.join(_HERE, "emu_harness.cpp")] + [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".h")]
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["hipcc", "-x", "hip", "--offload-host-only", "-O2", "-fPIC", "-shared", "-ffp-contract=off", *_DEFINES,
                               "-o", _LIB, os.path.join(_HERE, "emu_harness.cpp")])
    return _LIB


def run(cfgs, chain, step_chunk=0, pool_doubles=0, waves=1):
    """cfgs: list of resolved configs sharing one bucket. Returns dict of arrays [batch, ...]."""
    from robotic_mpc_amd import packing

    lib = C.CDLL(build())
    c0 = cfgs[0]
    B, N, Nsim = len(cfgs), c0["N"], c0["Nsim"]
    pb = Problem(B, N, Nsim, c0["solver_type"], c0["max_iter"], c0["qp_iter_max"], int(c0["fixed_step"]), 0)
    params = packing.pack_batch(cfgs)
    robot = np.ascontiguousarray(chain.packed(c0["t_ee"]))
    T1 = Nsim + 1
    o = dict(z=np.zeros((B, 12, T1)), u=np.zeros((B, 6, T1)), ee_pose=np.zeros((B, 12, T1)), ee_rpy=np.zeros((B, 3, T1)),
             ee_vel=np.zeros((B, 6, T1)), status=np.zeros((B, Nsim), np.int32), sqp_iter=np.zeros((B, Nsim), np.int32),
             qp_iter=np.zeros((B, Nsim), np.int32), residuals=np.zeros((B, Nsim, 4)), cost=np.zeros((B, Nsim)),
             solver_time=np.zeros((B, Nsim)), errors=np.zeros((B, 7, T1)), plant_time=np.zeros((B, Nsim)))
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    args = [C.byref(pb), robot.ctypes.data_as(dp), params.ctypes.data_as(dp)]
    for k in ("z", "u", "ee_pose", "ee_rpy", "ee_vel"):
        args.append(o[k].ctypes.data_as(dp))
    for k in ("status", "sqp_iter", "qp_iter"):
        args.append(o[k].ctypes.data_as(ip))
    for k in ("residuals", "cost", "solver_time", "errors", "plant_time"):
        args.append(o[k].ctypes.data_as(dp))
    args.append(C.c_int(step_chunk))
    args.append(C.c_int(pool_doubles))
    args.append(C.c_int(waves))
    rc = lib.emu_run(*args)
    assert rc == 0
    return o
