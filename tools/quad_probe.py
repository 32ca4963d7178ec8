"""Rollout kernel time of the four-row (tile_rows=4) against the sixteen-row kernel for a small MPPI
problem as a function of the horizon: separates the per-step cost from the fixed cost of a launch.
Usage: python tools/quad_probe.py [N] [hidden] [nx] [nu]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autompc_amd import MLP, _lib
from autompc_amd.system import System
from oracle import mlp as omlp

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
hid = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nx = int(sys.argv[3]) if len(sys.argv) > 3 else 2
nu = int(sys.argv[4]) if len(sys.argv) > 4 else 1
system = System(["x%d" % i for i in range(nx)], ["u%d" % i for i in range(nu)])
p = omlp.random_params(nx, nu, [hid, hid], "relu", seed=1)
m = MLP(system, n_hidden_layers=2, hidden_size=hid, nonlintype="relu")
m.weights, m.biases = p["weights"], p["biases"]
m.xu_means, m.xu_std, m.dy_means, m.dy_std = p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"]
h = _lib.Handle(0, "f64")
m.stage_into(h)
h.set_quad_costs(np.eye(nx), 0.01 * np.eye(nu), np.eye(nx), np.zeros(nx))
h.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
for rows in (16, 4):
    for H in (4, 30, 60):
        plan = _lib.MppiPlan(h, [N], [H], [0.25], [1.0])
        plan.set_geometry(rows, 0)
        plan.upload(x0=np.zeros((1, nx)), act_seq=np.zeros(H * nu))
        for _ in range(20):
            plan.generate_eps(1, 0); plan.solve()
        plan.set_timing(True)
        for i in range(200):
            plan.generate_eps(1, i); plan.solve()
        t = plan.timing()
        print("rows %2d  H %2d  rollout %.4f ms  update %.4f ms  (%d launches)"
              % (rows, H, t["rollout_ms"], t["update_ms"], t["count"]))
        plan.close()
