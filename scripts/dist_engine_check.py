#!/usr/bin/env python3
"""Sharded run_all on the HIP engine, N ranks, compared bit for bit with the single-process result.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
        scripts/dist_engine_check.py [--backend gloo|nccl]

One process per rank as on an 8-GPU node; with fewer GPUs than ranks (a 1-GPU box) the ranks share the card
and the gather runs over gloo (``--backend gloo``): the control flow -- bucket sharding, engine launches on every
rank, padded gather of the device results to rank 0 -- is the product path of SimulationManager.run_all; only the
transport differs from RCCL.  Rank 0 then runs the same queue unsharded and checks equality.  Prints one JSON line.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    import numpy as np
    import torch
    import torch.distributed as dist

    ndev = max(torch.cuda.device_count(), 1)
    local = int(os.environ.get("LOCAL_RANK", "0")) % ndev
    os.environ["LOCAL_RANK"] = str(local)
    torch.cuda.set_device(local)
    if args.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from robotic_mpc_amd import SimulationManager, base_params

    ok, detail = True, {}
    try:
        m = SimulationManager(base_params(prediction_horizon=12, simulation_time=0.2))
        m.grid_search({"prediction_horizon": [12, 30], "w_qddot": [0.02, 0.05, 0.08]},
                      surface_coeff_sets=[dict(a=-0.1, b=0.1, c=-0.01, d=0.01, e=0.01, f=0.0),
                                          dict(a=-0.12, b=0.09, c=-0.01, d=0.012, e=0.008, f=0.0)])   # 2 buckets of 6
        m.add_manual("odd", {"prediction_horizon": 7})                                                # bucket of 1 (< world)
        res = m.run_all(distributed=True)
        info = dict(m.last_run_info)
        if rank == 0:
            single = m.run_all(distributed=False)
            assert [r["name"] for r in res] == [r["name"] for r in single] and len(res) == 13
            for a, b in zip(res, single):
                for k in ("q", "qdot", "u", "ee_pose"):
                    assert np.array_equal(a["data"][k], b["data"][k]), (a["name"], k)
                for k in ("e1", "e2", "e3", "e4", "e5", "solver_status", "sqp_iterations"):
                    assert np.array_equal(a["analysis"][k], b["analysis"][k]), (a["name"], k)
                for k in ("weighted_rmse", "rmse_e1", "itse_e5", "total_sqp_iterations", "num_failures"):
                    assert a["summary"][k] == b["summary"][k], (a["name"], k)
            detail = {"n_sims": len(res), "world_size": info["world_size"], "buckets": info["buckets"],
                      "sharded_equals_unsharded": True, "backend": args.backend,
                      "devices": ndev, "weighted_rmse_first": res[0]["summary"]["weighted_rmse"]}
        else:
            assert res == []
    except Exception as e:   # report and exit non-zero WITHOUT entering another collective: the other rank may be gone
        print("DIST_ENGINE_CHECK " + json.dumps({"ok": False, "rank": rank, "error": repr(e)}), flush=True)
        sys.exit(1)
    dist.barrier()
    if rank == 0:
        print("DIST_ENGINE_CHECK " + json.dumps({"ok": ok, **detail}), flush=True)
    dist.destroy_process_group()
    sys.exit(0)


if __name__ == "__main__":
    main()
