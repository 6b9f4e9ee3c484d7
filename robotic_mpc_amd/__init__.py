"""MI355X-native batched MPC rollout engine: a drop-in for the hot path of
lynet55/robotic-mpc (SimulationManager.grid_search / sweep / run_all -> Simulator.run).

    from robotic_mpc_amd import SimulationManager, BASE_PARAMS
    mgr = SimulationManager(BASE_PARAMS)
    mgr.grid_search({"prediction_horizon": [50, 100], "w_qddot": [0.02, 0.05]})
    results = mgr.run_all()          # one batched GPU launch per (N, solver) bucket

See DESIGN.md for the path, its boundary and the kernels; INTEGRATION.md for the C ABI.
"""
__version__ = "0.1.0"

from .config import BASE_PARAMS, base_params, resolve_config  # noqa: E402,F401
from .simulator import SimulationManager, Simulator  # noqa: E402,F401

__all__ = ["SimulationManager", "Simulator", "BASE_PARAMS", "base_params", "resolve_config", "__version__"]
