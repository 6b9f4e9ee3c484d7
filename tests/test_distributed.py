"""N>1 path on CPU: two gloo ranks shard a queued grid search, gather on rank 0, and the result
equals the single-process run (sharded == unsharded, bit for bit)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist

    import helpers as hp
    from robotic_mpc_amd import SimulationManager, base_params

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = SimulationManager(base_params(prediction_horizon=5, simulation_time=0.06), runner=hp.oracle_runner)
        m.grid_search({"prediction_horizon": [4, 5], "w_qddot": [0.02, 0.05, 0.08]})   # buckets of 3 and 3
        m.add_manual("odd", {"prediction_horizon": 7})                                  # bucket of 1 (< world)
        res = m.run_all(distributed=True)
        if rank == 0:
            single = m.run_all(distributed=False)
            assert [r["name"] for r in res] == [r["name"] for r in single] and len(res) == 7
            for a, b in zip(res, single):
                for k in ("q", "qdot", "u", "ee_pose"):
                    assert np.array_equal(a["data"][k], b["data"][k]), (a["name"], k)
                assert a["summary"]["weighted_rmse"] == b["summary"]["weighted_rmse"]
            assert m.last_run_info["world_size"] == 1
            # summary-only gather: only the 24-scalar records cross the ranks; same summaries, no logs on the host
            open(os.path.join(outdir, "ok"), "w").write("ok")
        lite = m.run_all(distributed=True, results="summary")
        assert m.run_all(distributed=True, return_results=False) is None
        if rank == 0:
            assert m.last_run_info["results"] == "summary" and len(m.last_summaries) == 7
            for a, b, c in zip(lite, single, m.last_summaries):
                assert a["name"] == b["name"] == c["name"] and set(a.keys()) == {"name", "simulator", "summary"}
                for k, v in b["summary"].items():
                    if "time" not in k:      # (wall-clock fields differ run to run)
                        assert a["summary"][k] == v and c[k] == v, k
                assert a["simulator"].simulation_model is None
                with pytest.raises(RuntimeError, match="summary"):
                    a["data"]
            open(os.path.join(outdir, "ok_summary"), "w").write("ok")
        else:
            assert res == []
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_sharded_equals_single(tmp_path, orc):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists() and (tmp_path / "ok_summary").exists()


# ------------------------------------------------------------------------------------------- 8 ranks
def _worker8(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), OMP_NUM_THREADS="1")
    import torch
    import torch.distributed as dist

    torch.set_num_threads(1)
    import helpers as hp
    from robotic_mpc_amd import SimulationManager, base_params, distributed as dmod, packing

    class RaggedOracle:
        """oracle runner that also takes merged (ragged) buckets, like the engine runner does"""
        supports_ragged = True

        def __call__(self, cfgs, chain):
            return hp.oracle_runner(cfgs, chain)

    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = False
    try:
        old = packing.RAGGED_MIN_BATCH
        packing.RAGGED_MIN_BATCH = 2            # 3 horizons x 6 = 18 sims / 8 ranks >= 2 each: merged into ONE ragged bucket
        m = SimulationManager(base_params(prediction_horizon=4, simulation_time=0.03), runner=RaggedOracle())
        m.grid_search({"prediction_horizon": [3, 4, 6], "w_qddot": [0.02, 0.05, 0.08], "w_u": [0.01, 0.001]})   # 18 RTI sims
        m.add_manual("sqp_a", {"solver_options": {"nlp_solver_type": "SQP", "nlp_solver_max_iter": 3}})         # bucket of 2 (< world)
        m.add_manual("sqp_b", {"solver_options": {"nlp_solver_type": "SQP", "nlp_solver_max_iter": 3}, "w_u": 0.02})
        resolved = [__import__("robotic_mpc_amd").resolve_config(s["config"]) for s in m.simulations]
        buckets = dmod.group_buckets(resolved, world, True)
        keys = list(buckets)
        assert len(keys) == 2 and keys[0][0] == "ragged", keys
        ragged = buckets[keys[0]]
        assert sorted(ragged) == list(range(18))
        bounds = dmod.chunk_bounds(18, world)
        assert [hi - lo for lo, hi in bounds] == [3, 3, 2, 2, 2, 2, 2, 2] and bounds[0][0] == 0 and bounds[-1][1] == 18
        # every rank's contiguous shard of the ragged bucket starts with its longest horizon (dispatched first) and the
        # shards hold the same mix (round-robin deal of the horizon-sorted list)
        lo, hi = bounds[rank]
        mine = [resolved[i]["N"] for i in ragged[lo:hi]]
        assert mine == sorted(mine, reverse=True), mine
        assert dmod.chunk_bounds(2, world) == [(0, 1), (1, 2)] + [(2, 2)] * 6      # ranks 2..7 get nothing of the SQP bucket
        res = m.run_all(distributed=True)
        packing.RAGGED_MIN_BATCH = old
        if rank == 0:
            single = m.run_all(distributed=False)
            assert [r["name"] for r in res] == [r["name"] for r in single] and len(res) == 20
            for a, b in zip(res, single):
                for k in ("q", "qdot", "u", "ee_pose"):
                    assert np.array_equal(a["data"][k], b["data"][k]), (a["name"], k)
                assert a["summary"]["weighted_rmse"] == b["summary"]["weighted_rmse"]
                assert np.array_equal(a["simulator"].qp_iter, b["simulator"].qp_iter)
            open(os.path.join(outdir, "ok8"), "w").write("ok")
        else:
            assert res == []
        ok = True
    finally:
        if ok:                        # after a failure: no barrier (the other ranks may be gone) -- exit non-zero instead
            dist.barrier()
            dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_eight_rank_gloo_ragged_merge_and_small_buckets(tmp_path, orc):
    """The sharding the driver's 8-GPU run takes, on 8 gloo ranks: chunk bounds, one ragged bucket dealt over the ranks,
    a bucket smaller than the world (six ranks contribute empty shards to its gather), sharded == unsharded bit for bit."""
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker8, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    assert (tmp_path / "ok8").exists()


# ------------------------------------------------------------------------------------------- failure on one rank
def _worker_fail(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist

    import helpers as hp
    from robotic_mpc_amd import SimulationManager, base_params

    def runner(cfgs, chain):
        if rank == 1:
            raise RuntimeError("hand-off failed on this rank")      # what eng.sync() raises after a failed launch
        return hp.oracle_runner(cfgs, chain)

    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = SimulationManager(base_params(prediction_horizon=4, simulation_time=0.03), runner=runner)
    m.sweep("w_qddot", [0.02, 0.05, 0.08, 0.1])
    try:
        m.run_all(distributed=True)
    except RuntimeError as e:
        # rank 1 raises its own error, rank 0 is told by the all-reduce in front of the gather: nobody blocks in it
        open(os.path.join(outdir, f"raised{rank}"), "w").write(str(e))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_failure_on_one_rank_stops_every_rank_before_the_gather(tmp_path, orc):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker_fail, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert "hand-off failed" in (tmp_path / "raised1").read_text()
    assert "another rank failed" in (tmp_path / "raised0").read_text()
