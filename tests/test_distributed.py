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
            open(os.path.join(outdir, "ok"), "w").write("ok")
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
    assert (tmp_path / "ok").exists()
