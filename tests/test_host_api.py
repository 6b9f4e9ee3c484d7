"""Host layer: the reference's SimulationManager / Simulator API (simulator.py:15-716) and
post-run analysis, exercised without a GPU through an oracle-backed runner."""
import os
import warnings

import numpy as np
import pytest

import helpers as hp


def _base(**kw):
    from robotic_mpc_amd import base_params

    return base_params(prediction_horizon=6, simulation_time=0.08, **kw)


# ----------------------------------------------------------------------------- config
def test_config_validation_mirrors_simulator_signature():
    from robotic_mpc_amd import config

    cfg = config.base_params()
    r = config.resolve_config(cfg)
    assert r["Nsim"] == 600 and r["N"] == 100 and r["solver_type"] == config.SOLVER_RTI
    assert r["qp_tol"] == 1e-8 and r["tol"] == 1e-6 and r["max_iter"] == 100 and r["qp_iter_max"] == 50
    np.testing.assert_array_equal(r["coeffs"], [-0.15, 0.15, -0.01, 0.01, 0.01, 0.0])  # surface.py:14-17
    bad = dict(cfg)
    del bad["wcv"]  # required positional (simulator.py:22)
    with pytest.raises(TypeError, match="wcv"):
        config.resolve_config(bad)
    with pytest.raises(TypeError, match="unexpected"):
        config.resolve_config({**cfg, "not_a_param": 1})
    # default solver is SQP (trajectory_optimizer.py:60) when solver_options is absent
    d = dict(cfg)
    d.pop("solver_options")
    assert config.resolve_config(d)["solver_type"] == config.SOLVER_SQP
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        config.resolve_config({**cfg, "solver_options": {"nlp_solver_type": "SQP_RTI", "bogus": 3}})
        assert any("Unknown solver option 'bogus'" in str(x.message) for x in w)  # simulator.py:135
    with pytest.raises(ValueError):
        config.resolve_config({**cfg, "solver_options": {"nlp_solver_type": "DDP"}})
    # partial surface coefficients update the defaults (surface.py:18-19)
    r = config.resolve_config({**cfg, "surface_coeffs": {"a": -0.3}})
    assert r["coeffs"][0] == -0.3 and r["coeffs"][1] == 0.15


def test_result_changing_acados_options_are_refused_not_ignored():
    from robotic_mpc_amd import config

    """simulator.py:129-135 sets ANY attribute that exists on AcadosOcpOptions, so acados would honour
    alpha_min, full_step_dual, hpipm_mode ...; the engine implements their default only: a non-default value
    raises, the default is accepted silently, and names acados does not know keep the reference's warning."""
    cfg = _base()
    rti = {"nlp_solver_type": "SQP_RTI"}
    for key, bad in [("alpha_min", 0.01), ("alpha_reduction", 0.5), ("nlp_solver_step_length", 0.5),
                     ("line_search_use_sufficient_descent", 1), ("eps_sufficient_descent", 1e-2), ("full_step_dual", 1),
                     ("globalization_use_SOC", 1), ("hpipm_mode", "SPEED"), ("hpipm_mode", "ROBUST"), ("qp_solver_mu0", 10.0),
                     ("qp_solver_t0_init", 0), ("cost_discretization", "INTEGRATOR"), ("globalization_alpha_min", 0.2),
                     ("globalization_full_step_dual", 1), ("as_rti_level", 3), ("with_adaptive_levenberg_marquardt", True)]:
        with pytest.raises(ValueError, match=key):
            config.resolve_config({**cfg, "solver_options": {**rti, key: bad}})
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        r = config.resolve_config({**cfg, "solver_options": {**rti, "alpha_min": 0.05, "alpha_reduction": 0.7, "hpipm_mode": "BALANCE",
                                                             "full_step_dual": 0, "nlp_solver_step_length": 1.0, "qp_solver_mu0": 0}})
        assert not w and r["solver_type"] == config.SOLVER_RTI
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        config.resolve_config({**cfg, "solver_options": {**rti, "alpha_minimum": 0.01}})   # not an acados attribute
        assert any("Unknown solver option 'alpha_minimum'" in str(x.message) for x in w)


# ----------------------------------------------------------------------------- queueing
def test_sweep_and_grid_search_semantics():
    from robotic_mpc_amd import SimulationManager

    m = SimulationManager(_base())
    m.sweep("prediction_horizon", [50, 100], name_template="H={}")          # simulator.py:590
    m.sweep("surface_coeffs.a", [-0.2, -0.1])                               # :579-586
    assert [s["name"] for s in m.simulations] == ["H=50", "H=100", "surface_coeffs.a=-0.2", "surface_coeffs.a=-0.1"]
    assert m.simulations[2]["config"]["surface_coeffs"] == {"a": -0.2}
    assert m.simulations[3]["config"]["surface_coeffs"] == {"a": -0.1}
    assert "surface_coeffs" not in m.base_config or not m.base_config.get("surface_coeffs")
    m.clear()
    assert m.simulations == []
    sets = [{"a": -0.1, "b": 0.1, "c": 0, "d": 0, "e": 0, "f": 0}, {"a": -0.2, "b": 0.2, "c": 0, "d": 0, "e": 0, "f": 0}]
    m.grid_search({"prediction_horizon": [20, 50], "w_qddot": [0.02, 0.05]}, surface_coeff_sets=sets)
    names = [s["name"] for s in m.simulations]
    assert len(names) == 8 and names[0] == "coeffs0_prediction_horizon=20_w_qddot=0.02"     # :633-635
    assert names[-1] == "coeffs1_prediction_horizon=50_w_qddot=0.05"
    assert m.simulations[5]["config"]["surface_coeffs"] == sets[1] and m.simulations[5]["config"]["surface_coeffs"] is not sets[1]
    m.clear()
    m.grid_search({"w_u": [0.01, 0.001]})
    assert [s["name"] for s in m.simulations] == ["w_u=0.01", "w_u=0.001"]                  # :637
    m.clear()
    m.grid_search({"w_u": [0.01]}, name_template=lambda p, i: f"{i}:{p['w_u']}")           # :629
    m.grid_search({"w_u": [0.01]}, name_template=lambda p: f"one:{p['w_u']}")              # :630-631 fallback
    assert [s["name"] for s in m.simulations] == ["0:0.01", "one:0.01"]
    # dotted keys do not alias the base config's nested dict (deliberate fix of :614-618)
    m2 = SimulationManager(_base(surface_coeffs={"a": -0.15}))
    m2.grid_search({"surface_coeffs.a": [-0.3, -0.4]})
    assert m2.base_config["surface_coeffs"] == {"a": -0.15}
    assert [s["config"]["surface_coeffs"]["a"] for s in m2.simulations] == [-0.3, -0.4]


def test_readme_aliases():
    from robotic_mpc_amd import SimulationManager

    m = SimulationManager(_base())
    m.sweep_parameter("w_u", [0.01, 0.001])                                  # readme.md:66
    m.add_manual(name="custom", params={"w_qddot": 0.07, "surface_coeffs.b": 0.3})  # readme.md:69
    assert [s["name"] for s in m.simulations] == ["w_u=0.01", "w_u=0.001", "custom"]
    assert m.simulations[2]["config"]["w_qddot"] == 0.07 and m.simulations[2]["config"]["surface_coeffs"]["b"] == 0.3


# ----------------------------------------------------------------------------- analysis
def test_errors_match_literal_restatement_of_reference_loop():
    from robotic_mpc_amd import analysis

    rng = np.random.default_rng(0)
    T = 37
    R = np.stack([np.linalg.qr(rng.normal(size=(3, 3)))[0] for _ in range(T)])
    ee_pose = np.concatenate([rng.normal(size=(3, T)), R.reshape(T, 9).T])
    ee_vel = rng.normal(size=(6, T))
    coeffs = np.array([-0.15, 0.15, -0.01, 0.01, 0.01, 0.02])
    got = analysis.compute_errors(ee_pose, ee_vel, coeffs, [0, 0, 0.1], 0.4, 0.05)
    ref = hp.reference_errors_loop(ee_pose, ee_vel, coeffs, [0, 0, 0.1], 0.4, 0.05)
    assert set(got) == set(ref) == {"e1", "e2", "e3", "e4", "e5", "p_task_z", "p_ee_y"}   # simulator.py:344
    for k in ref:
        np.testing.assert_allclose(got[k], ref[k], atol=1e-14, err_msg=k)


def test_metrics_formulas():
    from robotic_mpc_amd import analysis

    rng = np.random.default_rng(1)
    errs = {k: rng.normal(size=11) for k in ("e1", "e2", "e3", "e4", "e5")}
    m = analysis.compute_metrics(errs, 0.01)
    t = np.arange(11) * 0.01
    assert abs(m["itse"]["e3"] - np.sum(t * errs["e3"] ** 2) * 0.01) < 1e-15             # simulator.py:365-369
    assert abs(m["rmse"]["e2"] - np.sqrt(np.mean(errs["e2"] ** 2))) < 1e-15              # :375-379
    w = np.sqrt(np.mean(sum(50.0 * errs[k] ** 2 for k in errs)))
    assert abs(m["weighted_rmse"] - w) < 1e-14                                           # :383-384


# ----------------------------------------------------------------------------- run_all
def test_run_all_result_schema_and_order(orc):
    from robotic_mpc_amd import SimulationManager, Simulator

    m = SimulationManager(_base(), runner=hp.oracle_runner)
    m.grid_search({"prediction_horizon": [4, 6], "w_u": [0.01, 0.001]})
    m.add_manual("sqp", {"solver_options": {"nlp_solver_type": "SQP"}})
    res = m.run_all()
    assert [r["name"] for r in res] == ["prediction_horizon=4_w_u=0.01", "prediction_horizon=4_w_u=0.001",
                                        "prediction_horizon=6_w_u=0.01", "prediction_horizon=6_w_u=0.001", "sqp"]
    assert m.last_run_info["buckets"] == 3
    r = res[2]
    assert set(r) == {"name", "simulator", "data", "analysis", "summary"}                  # simulator.py:668-674
    sim = r["simulator"]
    assert isinstance(sim, Simulator) and sim.name == r["name"] and sim.prediction_horizon == 6 and sim.dt == 0.01
    d = r["data"]
    assert set(d) == {"time", "q", "qdot", "qddot", "qddot_fd", "u", "ee_pose", "px_ref", "vy_ref", "N"}  # :477-488
    assert d["q"].shape == (6, 9) and d["ee_pose"].shape == (12, 9) and d["N"] == 6
    np.testing.assert_allclose(d["qddot"], -200.0 * d["qdot"] + 200.0 * d["u"])           # :466
    np.testing.assert_allclose(d["qddot_fd"][:, :-1], np.diff(d["qdot"], axis=1) / 0.01)  # :471
    np.testing.assert_array_equal(d["qddot_fd"][:, -1], d["qddot_fd"][:, -2])             # :474
    assert len(r["summary"]) == 20 and "weighted_rmse" in r["summary"]                    # :523-547
    for k in ("e1", "e5", "weighted_rmse", "rmse", "itse", "sqp_iterations", "kkt_residuals", "mpc_time",
              "computational_time_sim"):
        assert k in r["analysis"]
    # attributes plotter.py consumes (plotter.py:636-651, 766-777)
    assert sim.timings["mpc_time"].shape == (8,) and sim.errors["e1"].shape == (9,)
    # per-simulation results equal a standalone oracle run of the same config (order preserved)
    solo = hp.oracle_runner([sim.resolved], __import__("robotic_mpc_amd").robots.builtin_chain("ur10"))
    np.testing.assert_array_equal(sim.simulation_model.z, solo["z"][0])
    assert res[4]["simulator"].solver_stats["sqp_iterations"].max() >= 1


def test_accessing_results_before_run_raises():
    from robotic_mpc_amd import Simulator

    s = Simulator(**_base())
    for attr in ("errors", "metrics", "solver_stats", "timings"):
        with pytest.raises(RuntimeError, match="Must call run"):   # simulator.py:267-268
            getattr(s, attr)
    with pytest.raises(RuntimeError):
        s.get_summary()


def test_chunk_bounds_and_buckets():
    from robotic_mpc_amd import config, distributed as d

    assert d.chunk_bounds(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert d.chunk_bounds(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    cfgs = [config.resolve_config(config.base_params(prediction_horizon=n)) for n in (20, 50, 20, 100, 50)]
    b = d.group_buckets(cfgs)
    assert list(b.values()) == [[0, 2], [1, 4], [3]]


def test_plant_integrator_option():
    """simulation_model.py:13,39-51: integration_method in {Euler, RK2, RK3, RK4}, anything else raises ValueError."""
    from robotic_mpc_amd import config, packing

    for name, code in (("RK4", 0), ("Euler", 1), ("RK2", 2), ("RK3", 3)):
        r = config.resolve_config(config.base_params(integration_method=name))
        assert r["plant_integrator"] == code and packing.pack_params(r)[7] == code
    assert packing.pack_params(config.resolve_config(config.base_params()))[7] == 0   # simulator.py:85
    with pytest.raises(ValueError, match="Unknown integration method"):
        config.resolve_config(config.base_params(integration_method="RK5"))


def test_results_archive_round_trip_and_resume(orc, tmp_path):
    """on-disk result format (plain npz) and resume of a partially completed grid search."""
    from robotic_mpc_amd import SimulationManager, results_io

    calls = []

    def counting_runner(cfgs, chain):
        calls.append(len(cfgs))
        return hp.oracle_runner(cfgs, chain)

    ck = str(tmp_path / "grid.npz")
    base = _base()
    m = SimulationManager(base, runner=counting_runner)
    m.grid_search({"w_qddot": [0.02, 0.05]})
    first = m.run_all(checkpoint=ck)
    assert sum(calls) == 2 and os.path.exists(ck) and m.last_run_info["n_resumed"] == 0

    # a larger grid over the same file: only the new combinations run, results identical for the old ones
    calls.clear()
    m2 = SimulationManager(base, runner=counting_runner)
    m2.grid_search({"w_qddot": [0.02, 0.05, 0.1]})
    second = m2.run_all(checkpoint=ck)
    assert sum(calls) == 1 and m2.last_run_info["n_resumed"] == 2
    for a, b in zip(first, second[:2]):
        assert a["name"] == b["name"]
        np.testing.assert_array_equal(a["simulator"].simulation_model.z, b["simulator"].simulation_model.z)
        assert a["summary"] == b["summary"]

    # a changed config under an old name is NOT taken from the archive
    calls.clear()
    m3 = SimulationManager({**base, "px_ref": 0.45}, runner=counting_runner)
    m3.grid_search({"w_qddot": [0.02]})
    m3.run_all(checkpoint=str(tmp_path / "grid.npz"))
    assert sum(calls) == 1

    # the archive is the union of everything run so far and alone reproduces run_all's dicts
    loaded = results_io.load_results(ck)
    assert len(loaded) == 4 and {r["name"] for r in second} <= {r["name"] for r in loaded}
    by_name = {r["name"]: r for r in loaded if r["simulator"].px_ref == base["px_ref"]}
    assert by_name[first[0]["name"]]["summary"] == first[0]["summary"]
    arch = results_io.load_archive(ck)
    assert set(("names", "configs", "keys", "nsim", "z", "u", "status", "residuals")) <= set(arch)
    r0 = results_io.load_results(ck)[0]
    assert r0["summary"]["num_failures"] >= 0 and r0["simulator"].simulation_model.z.shape[0] == 12


def test_format_1_archive_and_11_array_runner_still_work(orc, tmp_path):
    """an archive written before `errors` / `plant_time` were logged (format 1) still resumes -- the task errors
    are recomputed from its logs -- and a custom runner written to the 11-array contract is completed, not a KeyError."""
    from robotic_mpc_amd import SimulationManager, results_io

    def old_runner(cfgs, chain):
        out = hp.oracle_runner(cfgs, chain)
        out.pop("errors"); out.pop("plant_time")
        return out

    base = _base()
    m = SimulationManager(base, runner=old_runner)
    m.grid_search({"w_qddot": [0.02, 0.05]})
    res_old = m.run_all()
    m_new = SimulationManager(base, runner=hp.oracle_runner)
    m_new.grid_search({"w_qddot": [0.02, 0.05]})
    res_new = m_new.run_all()
    for a, b in zip(res_old, res_new):
        for k in ("rmse_e1", "itse_e5", "weighted_rmse", "total_sqp_iterations", "max_kkt_residual"):   # (timings differ run to run)
            assert a["summary"][k] == b["summary"][k]
        for k in ("e1", "e5", "p_ee_y"):
            np.testing.assert_array_equal(a["analysis"][k], b["analysis"][k])
    with pytest.raises(KeyError, match="ee_vel"):
        _queued(base, lambda c, ch: {k: v for k, v in hp.oracle_runner(c, ch).items() if k != "ee_vel"}).run_all()

    ck = str(tmp_path / "v2.npz")
    m_new.run_all(checkpoint=ck)
    arch = dict(np.load(ck, allow_pickle=False))
    v1 = {k: v for k, v in arch.items() if k not in ("errors", "plant_time")}
    v1["format_version"] = np.int64(1)
    ck1 = str(tmp_path / "v1.npz")
    np.savez_compressed(ck1, **v1)
    back = results_io.load_archive(ck1)
    np.testing.assert_allclose(back["errors"], arch["errors"], rtol=0, atol=0)
    assert not back["plant_time"].any()
    calls = []
    m3 = SimulationManager(base, runner=lambda c, ch: calls.append(len(c)) or hp.oracle_runner(c, ch))
    m3.grid_search({"w_qddot": [0.02, 0.05, 0.1]})
    out = m3.run_all(checkpoint=ck1)                 # resumes from the v1 file: only the new combination runs
    assert sum(calls) == 1 and m3.last_run_info["n_resumed"] == 2
    assert out[0]["summary"]["weighted_rmse"] == res_new[0]["summary"]["weighted_rmse"]
    with pytest.raises(ValueError, match="unsupported format version 3"):
        v1["format_version"] = np.int64(3)
        np.savez_compressed(ck1, **v1)
        results_io.load_archive(ck1)


def _queued(base, runner):
    from robotic_mpc_amd import SimulationManager

    m = SimulationManager(base, runner=runner)
    m.sweep("w_qddot", [0.02])
    return m


def test_runner_with_submit_collect_launches_all_buckets_first(orc):
    """distributed.run_partitioned: a runner that offers submit()/collect() gets every bucket submitted before the
    first collect (small buckets overlap on the GPU); results still come back in queue order."""
    from robotic_mpc_amd import SimulationManager

    log = []

    class AsyncRunner:
        def __call__(self, cfgs, chain):
            raise AssertionError("the synchronous path must not be used when submit() exists")

        def submit(self, cfgs, chain):
            log.append(("submit", cfgs[0]["N"], len(cfgs)))
            return cfgs, chain

        def collect(self, ticket):
            log.append(("collect", ticket[0][0]["N"]))
            return hp.oracle_runner(*ticket)

    m = SimulationManager(_base(), runner=AsyncRunner())
    m.grid_search({"prediction_horizon": [4, 6], "w_qddot": [0.02, 0.05]})
    res = m.run_all()
    assert [e[0] for e in log] == ["submit", "submit", "collect", "collect"]
    assert [r["simulator"].prediction_horizon for r in res] == [4, 4, 6, 6]
    ref = SimulationManager(_base(), runner=hp.oracle_runner)
    ref.grid_search({"prediction_horizon": [4, 6], "w_qddot": [0.02, 0.05]})
    for a, b in zip(res, ref.run_all()):
        np.testing.assert_array_equal(a["simulator"].simulation_model.z, b["simulator"].simulation_model.z)


# ----------------------------------------------------------------------------- options that change results
def test_solver_options_that_change_results_are_honoured_or_refused():
    """simulator.py:129-135 forwards ANY acados option by setattr: the ones that change results reach the
    parameter record, the ones the engine cannot honour raise -- nothing result-changing is ignored."""
    from robotic_mpc_amd import config, packing

    cfg = config.base_params()
    so = lambda **kw: config.resolve_config({**cfg, "solver_options": {"nlp_solver_type": "SQP", **kw}})
    r = so(tol=1e-5)                                            # the acados `tol` setter writes all four
    assert (r["tol"], r["tol_eq"], r["tol_ineq"], r["tol_comp"]) == (1e-5,) * 4
    r = so(tol=1e-5, nlp_solver_tol_stat=1e-3)                  # dict order = setattr order
    assert r["tol"] == 1e-3 and r["tol_eq"] == 1e-5
    r = so(nlp_solver_tol_stat=1e-3, tol=1e-5)
    assert r["tol"] == 1e-5
    r = so(nlp_solver_tol_comp=1e-4, levenberg_marquardt=1e-3)
    p = packing.pack_params(r)
    assert p[1] == 1e-6 and p[63] == 1e-4 and p[61] == 1e-6 and p[64] == 1e-3
    for bad in (dict(levenberg_marquardt=-1.0), dict(qp_solver="FULL_CONDENSING_QPOASES"), dict(qp_solver_warm_start=0),
                dict(qp_solver_tol_stat=1e-6), dict(regularize_method="MIRROR"), dict(tol=0.0)):
        with pytest.raises(ValueError):
            so(**bad)
    so(qp_solver="FULL_CONDENSING_HPIPM", qp_solver_cond_N=5, qp_solver_tol_stat=1e-8)   # same QP solution to qp_tol


def test_oracle_levenberg_marquardt_and_tolerances(orc, ur10_rb):
    """The oracle side of the options above: the LM term regularises the step (smaller first move), looser NLP
    tolerances stop the SQP loop earlier."""
    from robotic_mpc_amd import config

    mk = lambda **kw: config.resolve_config(config.base_params(prediction_horizon=8, simulation_time=0.05,
                                                               solver_options={"nlp_solver_type": "SQP", **kw}))
    plain = orc.run(ur10_rb, orc.make_params(mk()))
    lm = orc.run(ur10_rb, orc.make_params(mk(levenberg_marquardt=5.0)))
    loose = orc.run(ur10_rb, orc.make_params(mk(tol=1e-2)))
    assert np.abs(lm["u"] - plain["u"]).max() > 1e-4
    assert lm["sqp_iter"].sum() > plain["sqp_iter"].sum()       # damped steps: more SQP iterations to the same tolerance
    assert loose["sqp_iter"].sum() < plain["sqp_iter"].sum()
    assert (plain["status"] == 0).all() and (loose["status"] == 0).all()


def test_batch_summary_equals_per_simulation_analysis(orc):
    """analysis.batch_summary (the host mirror of the mpcb_summary kernel) against the per-simulation formulas of
    simulator.py:347-448 applied one simulation at a time."""
    from robotic_mpc_amd import SimulationManager, analysis
    from robotic_mpc_amd.engine import SUMMARY_COLS

    m = SimulationManager(_base(), runner=hp.oracle_runner)
    m.grid_search({"w_qddot": [0.02, 0.05, 0.1]})
    res = m.run_all()
    for r in res:
        sim = r["simulator"]
        assert sim._summary_row is not None
        fast = r["summary"]
        sim._summary_row = None                                   # force the per-simulation path
        sim._invalidate_cache(); sim._data_computed = True
        slow = sim.get_summary()
        assert list(fast) == list(slow) == list(SUMMARY_COLS[:20])
        for k in slow:
            assert fast[k] == pytest.approx(slow[k], rel=1e-12, abs=1e-15), k
        m2 = analysis.compute_metrics(sim.errors, sim.dt)
        assert r["analysis"]["rmse"]["e3"] == pytest.approx(m2["rmse"]["e3"], rel=1e-12)


def test_result_items_are_lazy_dicts(orc):
    from robotic_mpc_amd import SimulationManager

    m = SimulationManager(_base(), runner=hp.oracle_runner)
    m.sweep("w_u", [0.01, 0.001])
    r = m.run_all()[0]
    assert isinstance(r, dict) and "simulator" in r and "data" in r and len(r) == 5
    assert not dict.__contains__(r, "data")                       # nothing derived yet
    assert r["data"]["q"].shape == (6, 9) and dict.__contains__(r, "data")
    assert sorted(r.keys()) == ["analysis", "data", "name", "simulator", "summary"]
    assert r.get("nope", 7) == 7
    with pytest.raises(KeyError):
        r["nope"]


@pytest.mark.skipif(not os.path.exists("/root/reference/plotter.py"), reason="reference checkout not present (GPU box)")
def test_reference_plotter_consumes_run_all_results(orc):
    """the reference's own report code (plotter.py:608-737 box plot of sim.timings[...],
    :740-877 error envelope of sim.errors[...]) runs unchanged on what run_all returns.  The reference module is
    imported from /root/reference in this container only; nothing of it is shipped."""
    import importlib.util

    import matplotlib
    matplotlib.use("Agg")
    spec = importlib.util.spec_from_file_location("ref_plotter", "/root/reference/plotter.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from robotic_mpc_amd import SimulationManager

    def runner(cfgs, chain):      # the oracle delivers no plant time; give the plot something positive to draw
        out = hp.oracle_runner(cfgs, chain)
        out["plant_time"] = np.full_like(out["solver_time"], 2e-6)
        return out

    m = SimulationManager(_base(), runner=runner)
    m.grid_search({"prediction_horizon": [4, 6, 8]}, surface_coeff_sets=[dict(a=-0.1, b=0.1, c=-0.01, d=0.01, e=0.01, f=0.0),
                                                                        dict(a=-0.12, b=0.1, c=-0.01, d=0.01, e=0.01, f=0.0)])
    res = m.run_all()
    pl = mod.Plotter()
    for src in ("mpc_time", "integration_time", "solver_time", "total_computation_time"):
        fig, ax = pl.fig6_computation_time_boxplot_mpl(res, time_source=src)
        labels = [t.get_text() for t in ax.get_xticklabels()]
        assert len(ax.patches) + len(ax.lines) > 0 and len(labels) >= 3, src
        matplotlib.pyplot.close(fig)
    for key in ("e1", "e2", "e3", "e4", "e5"):
        fig = pl.error_envelope(res, error_key=key, band="std", show_individual=2)
        assert len(fig.data) >= 2
        mean = [tr for tr in fig.data if tr.y is not None and len(tr.y) == 9]
        assert mean, key
    fig = pl.error_envelope([r["simulator"] for r in res], error_key="e4", band="minmax")
    assert len(fig.data) >= 2
    # the integration-time box is no longer empty
    assert (res[0]["simulator"].timings["integration_time"] > 0).all()


def test_riccati_precision_is_a_bucket_property():
    """BASELINE configs[4]: `riccati_precision` travels config -> bucket key -> mpcb_problem.precision."""
    from robotic_mpc_amd import config, distributed, engine, packing

    a = config.resolve_config(config.base_params())
    b = config.resolve_config(config.base_params(riccati_precision="fp32"))
    assert a["precision"] == 0 and b["precision"] == 1
    assert packing.bucket_key(a) != packing.bucket_key(b)
    assert len(distributed.group_buckets([a, b, a])) == 2
    assert engine.make_problem([b]).precision == 1 and engine.make_problem([a, a]).precision == 0
    with pytest.raises(ValueError):
        engine.make_problem([a, b])
    with pytest.raises(ValueError):
        config.resolve_config(config.base_params(riccati_precision="fp16"))
    with pytest.raises(ValueError, match="SQP_RTI"):
        config.resolve_config(config.base_params(riccati_precision="fp32", solver_options={"nlp_solver_type": "SQP"}))


def test_reference_dump_comparer(orc, tmp_path):
    """compare.py: the array comparison behind `python -m robotic_mpc_amd.compare ref_run.npz` (the reference-run
    dump format of INTEGRATION.md), exercised with an oracle run standing in for both sides."""
    from robotic_mpc_amd import SimulationManager, compare

    m = SimulationManager(_base(), runner=hp.oracle_runner)
    m.sweep("w_u", [0.01])
    r = m.run_all()[0]
    sim, d = r["simulator"], r["data"]
    ref = {"q": d["q"], "qdot": d["qdot"], "u": d["u"], **{k: sim.errors[k] for k in ("e1", "e2", "e3", "e4", "e5")}}
    rep = compare.compare_arrays(ref, ref, 1e-6, sim.solver_status, sim.solver_status)
    assert rep["ok"] and rep["columns_compared"] == 9 and max(rep["max_abs_diff"].values()) == 0.0
    off = {k: v.copy() for k, v in ref.items()}
    off["u"][2, 5] += 3e-6
    rep = compare.compare_arrays(ref, off, 1e-6)
    assert not rep["ok"] and rep["first_violation"] == {"u": 5} and rep["max_abs_diff"]["u"] == pytest.approx(3e-6)
    st = np.zeros(8, dtype=int); st[3] = 2                                  # a flagged step ends the strict comparison
    rep = compare.compare_arrays(ref, off, 1e-6, st, None)
    assert rep["ok"] and rep["columns_compared"] == 4
    # the dump format round-trips through npz
    import json
    cfg = {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in _base().items()}
    np.savez(tmp_path / "ref_run.npz", config=json.dumps(cfg), **ref, solver_status=sim.solver_status)
    with np.load(tmp_path / "ref_run.npz") as f:
        back = compare._config_from_npz(f)
    assert back["prediction_horizon"] == _base()["prediction_horizon"] and np.allclose(back["q_0"], _base()["q_0"])


def test_ragged_buckets_merge_rti_horizons(orc, monkeypatch):
    """SQP_RTI buckets that differ only in prediction_horizon merge into one ragged launch once every rank gets
    RAGGED_MIN_BATCH simulations of it; each contiguous shard holds the same horizon mix, longest first; other solver
    types and small queues keep one bucket per horizon; results come back in queue order either way."""
    from robotic_mpc_amd import SimulationManager, config, distributed, packing

    mk = lambda N, **kw: config.resolve_config({**_base(), "prediction_horizon": N, **kw})
    res = [mk(N) for N in (4, 6, 8) for _ in range(4)] + [mk(6, solver_options={"nlp_solver_type": "SQP"}) for _ in range(3)]
    assert len(distributed.group_buckets(res)) == 4                                    # below the threshold: per horizon
    monkeypatch.setattr(packing, "RAGGED_MIN_BATCH", 5)
    b = distributed.group_buckets(res, parts=2)
    keys = list(b)
    assert len(b) == 2 and keys[0][0] == "ragged" and len(b[keys[0]]) == 12 and len(b[keys[1]]) == 3
    order = [res[i]["N"] for i in b[keys[0]]]
    assert order[:6] == [8, 8, 6, 6, 4, 4] and order[6:] == [8, 8, 6, 6, 4, 4]         # two shards, same mix, longest first
    assert len(distributed.group_buckets(res, parts=4)) == 4                           # 12 / 4 ranks < 5: no merge
    assert len(distributed.group_buckets(res, parts=2, merge_ragged=False)) == 4
    # end to end through run_all with a runner that accepts ragged buckets (the oracle runs every config by itself)
    calls = []

    class Runner:
        supports_ragged = True

        def __call__(self, cfgs, chain):
            calls.append(sorted({c["N"] for c in cfgs}))
            return hp.oracle_runner(cfgs, chain)

    m = SimulationManager(_base(), runner=Runner())
    m.grid_search({"prediction_horizon": [4, 6, 8], "w_qddot": [0.02, 0.05]})
    out = m.run_all()
    assert calls == [[4, 6, 8]] and m.last_run_info["buckets"] == 1
    ref = SimulationManager(_base(), runner=hp.oracle_runner)
    ref.grid_search({"prediction_horizon": [4, 6, 8], "w_qddot": [0.02, 0.05]})
    for a, r in zip(out, ref.run_all()):
        assert a["name"] == r["name"] and a["simulator"].prediction_horizon == r["simulator"].prediction_horizon
        np.testing.assert_array_equal(a["data"]["u"], r["data"]["u"])
    assert ref.last_run_info["buckets"] == 3


def test_ragged_problem_construction():
    from robotic_mpc_amd import config, engine, packing

    a = config.resolve_config({**_base(), "prediction_horizon": 5})
    b = config.resolve_config({**_base(), "prediction_horizon": 9})
    pb = engine.make_problem([a, b, a])
    assert pb.N == 9 and pb.batch == 3
    assert packing.pack_params(a)[65] == 5 and packing.pack_params(b)[65] == 9
    s = config.resolve_config({**_base(), "prediction_horizon": 9, "solver_options": {"nlp_solver_type": "SQP"}})
    s5 = config.resolve_config({**_base(), "prediction_horizon": 5, "solver_options": {"nlp_solver_type": "SQP"}})
    with pytest.raises(ValueError, match="SQP_RTI"):
        engine.make_problem([s, s5])


def test_run_all_summary_only_mode_moves_the_summaries_alone(ur10):
    """with results='summary' (or return_results=False) only the per-simulation summary record leaves the runner's
    buffers -- the analogue of gathering 192 B instead of ~262 KB per simulation across the GPUs; simulator.py:641-676 keeps the
    list shape, the items carry name / simulator / summary."""
    import helpers as hp
    from robotic_mpc_amd import SimulationManager, base_params, distributed as dmod

    m = SimulationManager(base_params(prediction_horizon=4, simulation_time=0.05), runner=hp.oracle_runner)
    m.sweep("w_qddot", [0.02, 0.05, 0.08])
    full = m.run_all()
    lite = m.run_all(results="summary")
    recs = dmod.run_partitioned([r["simulator"].resolved for r in full], hp.oracle_runner, lambda c: ur10, False, only=("summary",))
    assert all(set(r) == {"summary"} and r["summary"].shape == (24,) for r in recs)     # nothing with a time axis comes back
    assert [r["name"] for r in lite] == [r["name"] for r in full]
    for a, b in zip(lite, full):
        same = lambda x, y: all(x[k] == y[k] for k in x if "time" not in k) and set(x) == set(y)     # (wall-clock fields differ run to run)
        assert same(a["summary"], b["summary"]) and len(a) == 3 and "data" not in a and a["simulator"].simulation_model is None
        assert a["simulator"].metrics["weighted_rmse"] == b["summary"]["weighted_rmse"]
        with pytest.raises(RuntimeError, match="results='summary'"):
            a["simulator"].errors
    assert m.run_all(return_results=False) is None and m.last_run_info["results"] == "summary"
    assert [s["weighted_rmse"] for s in m.last_summaries] == [r["summary"]["weighted_rmse"] for r in full]
    with pytest.raises(ValueError):
        m.run_all(results="summary", checkpoint="/tmp/never_written.npz")
