"""Per-launch time of the iLQR line search on a full batch: B never-converging HalfCheetah problems (every
slot active in every iteration), per-iteration kernel times from the plan's HIP events.
  python tools/ls_rate.py [B] [iters]       AMPC_LS4_RB=1: four-row passes, =3 (default): one twelve-row pass"""
import os
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from autompc_amd import _lib                                   # noqa: E402
from autompc_amd.synthetic import make_workload                # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
system, task, model, spec = make_workload("c3", precision="f64", device=0)
nx, nu = spec["nx"], spec["nu"]
Q, R, F = task.get_cost().get_cost_matrices()
h = _lib.Handle(0, "f64")
model.stage_into(h)
h.set_quad_costs(Q, R, F, task.get_cost().get_goal())
h.set_ctrl_bounds(np.full(nu, -0.25), np.full(nu, 0.25))
rng = np.random.default_rng(0)
x0 = rng.uniform(-0.1, 0.1, size=(B, nx))
plan = _lib.IlqrPlan(h, B, 50, system.dt, clip_to_bounds=True)
plan.solve(x0, np.zeros((B, 50, nu)), max_iter=3)
plan.set_timing(True)
o = plan.solve(x0, np.zeros((B, 50, nu)), max_iter=iters)
t = plan.timing()
st = plan.stats()
print("B %d, %d iterations, AMPC_LS4_RB=%s AMPC_LIB=%s: rows/iter/problem %.2f  %s" % (
    B, iters, os.environ.get("AMPC_LS4_RB", "-"), os.path.basename(os.environ.get("AMPC_LIB", "product")),
    st["candidate_rows"] / max(1, int(o["iters"].sum())),
    {k: round(float(v), 4) for k, v in t.items()}))
print("   checksum ctrls %.17g  obj %.17g" % (float(np.abs(o["ctrls"]).sum()), float(o["objective"].sum()) if "objective" in o else 0.0))
