"""MI355X-native batched MPC rollout engine (drop-in for lynet55/robotic-mpc's
SimulationManager.grid_search / run_all hot path).  See DESIGN.md."""
__version__ = "0.1.0"
