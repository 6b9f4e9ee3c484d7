"""Build libmpcbatch.so (HIP, gfx950 only) in-tree with hipcc."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmpcbatch.so")
LIB_PROF = os.path.join(HERE, "libmpcbatch_prof.so")
ARCH = "gfx950"


def _stale(target: str) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(HERE), "include", "mpcbatch.h")]
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force: bool = False, profile: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> robotic_mpc_amd/libmpcbatch.so (cross-compiles without a GPU)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    target = LIB_PROF if profile else LIB
    if not force and not _stale(target):
        return target
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-fPIC", "-shared", "-std=c++17", "-o", target,
           os.path.join(CSRC, "mpc_kernel.hip")]
    if profile:
        cmd.insert(1, "-DMPCB_PROFILE")
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return target


def build_variant(name: str, defines) -> str:
    """Diagnostic builds (scripts/): robotic_mpc_amd/libmpcbatch_<name>.so with extra -D switches, rebuilt when stale."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    target = os.path.join(HERE, f"libmpcbatch_{name}.so")
    if _stale(target):
        subprocess.check_call([hipcc, f"--offload-arch={ARCH}", "-O3", "-fPIC", "-shared", "-std=c++17", *[f"-D{d}" for d in defines],
                               "-o", target, os.path.join(CSRC, "mpc_kernel.hip")])
    return target


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, profile="--profile" in sys.argv, verbose=True))
