"""The C-ABI library builds for gfx950 here (no GPU), loads, exports every symbol that
include/mpcbatch.h declares, and refuses loudly to run without a device."""
import ctypes as C
import sys
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from robotic_mpc_amd import build, engine

    build.build()
    return engine.load_library()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "mpcbatch.h")).read()
    body = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(mpcb_[a-z_]+)\s*\(", body)))
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"include/mpcbatch.h declares {n} but libmpcbatch.so does not export it"


def test_version_and_sizes(lib):
    from robotic_mpc_amd import engine, packing

    assert lib.mpcb_version() == 200
    pb = engine.MpcbProblem(256, 100, 600, 1, 100, 50, 0, 0)
    ws = lib.mpcb_workspace_bytes(C.byref(pb))
    assert 256 * 101 * 600 * 8 < ws < 256 * 101 * 1200 * 8  # a few hundred doubles per stage
    per_sim = lib.mpcb_result_bytes_per_sim(C.byref(pb))
    assert per_sim == (46 * 601 + 7 * 600) * 8 + 3 * 600 * 4
    assert packing.NPARAM == 72
    bad = engine.MpcbProblem(0, 100, 600, 1, 100, 50, 0, 0)
    assert lib.mpcb_workspace_bytes(C.byref(bad)) == 0


def test_engine_selection_thresholds(lib, monkeypatch):
    """mpcb_engine_for (host logic, no device): which kernel family a uniform bucket goes to, at both sides of every measured
    crossover of include/mpcbatch.h (profiles/r04_engine_sweep2.txt, r04_engine_sweep3.txt), and the MPCB_ENGINE override."""
    from robotic_mpc_amd import engine

    monkeypatch.delenv("MPCB_ENGINE", raising=False)
    hdr = open(os.path.join(ROOT, "include", "mpcbatch.h")).read()
    rti, sqp, steps = (int(re.search(r"#define %s\s+(\d+)" % n, hdr).group(1))
                       for n in ("MPCB_STREAM_MIN_BATCH", "MPCB_STREAM_MIN_BATCH_SQP", "MPCB_STREAM_MIN_STEPS_SQP"))
    assert (rti, sqp, steps) == (1280, 3072, 300)
    for batch, N, nsim, solver, prec, want in (
            (256, 100, 600, "SQP_RTI", 0, 0), (rti - 1, 100, 600, "SQP_RTI", 0, 0), (rti, 100, 600, "SQP_RTI", 0, 1),
            (4096, 100, 600, "SQP_RTI", 0, 1), (64, 300, 150, "SQP_RTI", 1, 1),          # fp32 Riccati: throughput engine only
            (2048, 100, 600, "SQP", 0, 0), (sqp - 1, 100, 600, "SQP", 0, 0), (sqp, 100, 600, "SQP", 0, 1),
            (4096, 100, steps - 1, "SQP", 0, 0), (4096, 100, steps, "SQP", 0, 1)):
        assert engine.engine_for(batch, N, nsim, solver, prec, lib=lib) == want, (batch, N, nsim, solver, prec)
    monkeypatch.setenv("MPCB_ENGINE", "stream")
    assert engine.engine_for(8, 100, 600, "SQP", lib=lib) == 1
    monkeypatch.setenv("MPCB_ENGINE", "latency")
    assert engine.engine_for(4096, 100, 600, "SQP_RTI", lib=lib) == 0
    bad = engine.MpcbProblem(0, 100, 600, 1, 100, 50, 0, 0)
    assert lib.mpcb_engine_for(C.byref(bad)) < 0


def test_struct_layouts_match_header():
    from robotic_mpc_amd import engine

    assert C.sizeof(engine.MpcbProblem) == 32
    assert C.sizeof(engine.MpcbResult) == 13 * 8


@pytest.mark.skipif(__import__("conftest").has_gpu(), reason="checks the no-device behaviour")
def test_no_device_fails_loudly(lib):
    from robotic_mpc_amd import engine

    assert lib.mpcb_device_count() == 0
    h = C.c_void_p()
    assert lib.mpcb_create(C.byref(h), 0) == -2  # MPCB_ENODEV
    assert not h.value
    with pytest.raises(engine.EngineError, match="no CPU fallback"):
        engine.MpcBatchEngine(0)
    assert lib.mpcb_last_error(None) == b"invalid handle"


def test_missing_library_fails_loudly(tmp_path):
    from robotic_mpc_amd import engine

    with pytest.raises(engine.EngineError, match="no CPU fallback"):
        engine.load_library(str(tmp_path / "libmpcbatch.so"))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under robotic_mpc_amd/ may reference it."""
    pkg = os.path.join(ROOT, "robotic_mpc_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "from oracle" not in txt and "import oracle" not in txt and "libmpc_oracle" not in txt, f
                assert "mpc_oracle.h" not in txt, f


def test_param_packing_layout():
    from robotic_mpc_amd import config, packing

    cfg = config.resolve_config(config.base_params())
    p = packing.pack_params(cfg)
    assert p.shape == (72,) and p[0] == 0.01 and p[2] == 1e-8 and p[1] == 1e-6
    assert p[61] == p[62] == p[63] == 1e-6 and p[64] == 0.0
    np.testing.assert_array_equal(p[8:14], [200.0] * 6)
    np.testing.assert_array_equal(p[14:20], cfg["q0"])
    np.testing.assert_array_equal(p[38:44], cfg["umin"])
    np.testing.assert_array_equal(p[50:56], [-0.15, 0.15, -0.01, 0.01, 0.01, 0.0])
    np.testing.assert_array_equal(p[56:61], [50.0] * 5)


def test_asm_scanner_detects_empty_exec_reload():
    """scripts/check_asm.py: the pattern of the hipcc 7.2 miscompile is recognised."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("check_asm", os.path.join(ROOT, "scripts", "check_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = "\n".join(["_Z3foov:", ".LBB0_1:", "\ts_andn2_b64 exec, exec, s[8:9]", "\ts_cbranch_execnz .LBB0_1",
                     "\tv_accvgpr_read_b32 v61, a11", "\ts_barrier", "\ts_branch .LBB0_2"])
    good = "\n".join(["_Z3foov:", ".LBB0_1:", "\ts_andn2_b64 exec, exec, s[8:9]", "\ts_cbranch_execnz .LBB0_1",
                      "\ts_or_b64 exec, exec, s[8:9]", "\tv_accvgpr_read_b32 v61, a11"])
    assert len(mod.scan(bad)) == 1 and mod.scan(bad)[0][0] == "_Z3foov"
    assert mod.scan(good) == []
    # second check: register spills in the item-parallel passes (a scratch reload drains every prefetch in flight)
    hot = "_ZN4mpcb6EngineI7DevExecILi8ELi1EEE12fwd_residentILb0ELb0EEEdv"
    spilled = "\n".join([hot + ":"] + ["\tscratch_load_dword v1, off, s32"] * 12 + [".Lfunc_end0:", "_Z5otherv:", "\tscratch_load_dword v1, off, s32"])
    assert mod.scratch_ops(spilled) == {hot: (12, mod.MAX_SCRATCH_OPS)}           # name -> (count, limit)
    # round 4: the two-per-CU build and the throughput engine's sweeps are guarded too; functions with a recorded budget pass below it
    for name in ("_ZN4mpcb6EngineI7DevExecILi4ELi2EEE15residual_directEid", "_ZN4mpcb2se9fact_passIdLb0EEEvv"):
        txt = "\n".join([name + ":"] + ["\tscratch_load_dword v1, off, s32"] * 9 + [".Lfunc_end0:"])
        assert mod.scratch_ops(txt) == {name: (9, mod.MAX_SCRATCH_OPS)}, name
    nlp = "_ZN4mpcb6EngineI7DevExecILi8ELi1EEE10nlp_directEdbbPdb"
    assert mod.scratch_ops("\n".join([nlp + ":"] + ["\tscratch_load_dword v1, off, s32"] * 40 + [".Lfunc_end0:"])) == {}     # (budget 46 since the joint-angle sincos)
    assert nlp in mod.scratch_ops("\n".join([nlp + ":"] + ["\tscratch_load_dword v1, off, s32"] * 200 + [".Lfunc_end0:"]))


def test_generated_isa_has_no_vector_op_under_empty_exec():
    """The shipped kernel source compiles (gfx950) without the miscompile pattern and without register spills in the hot
    item-parallel / recursion passes of the 4- and 8-wavefront builds."""
    import subprocess

    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_asm.py")], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
