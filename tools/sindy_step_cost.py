"""Rollout-kernel time of the CartPole SINDy MPPI solve (BASELINE config 1's model, 256 samples) over horizons:
the slope is the cost of a time step, the intercept the fixed cost of a launch.
python tools/sindy_step_cost.py     (AMPC_SINDY_FP=0: one thread per sample; AMPC_SINDY_G=16|32|64)"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from autompc_amd import _lib                                   # noqa: E402
from autompc_amd.synthetic import make_workload                # noqa: E402

system, task, model, spec = make_workload("c1", precision="f64", device=0)
nx, nu = spec["nx"], spec["nu"]
Q, R, F = task.get_cost().get_cost_matrices()
h = _lib.Handle(0, "f64")
model.stage_into(h)
h.set_quad_costs(Q, R, F, np.zeros(nx))
h.set_ctrl_bounds(np.full(nu, -20.0), np.full(nu, 20.0))
res = []
for H in (5, 10, 20, 40, 80):
    plan = _lib.MppiPlan(h, [256], [H], [1.0], [1.0])
    plan.upload(np.tile(task.get_init_obs(), (1, 1)), np.zeros(H * nu))
    plan.set_outputs(keep_eps_out=False)
    for i in range(50):
        plan.generate_eps(1, i); plan.solve()
    plan.set_timing(True)
    for i in range(300):
        plan.generate_eps(1, 50 + i); plan.solve()
    t = plan.timing()
    res.append((H, t["rollout_ms"], t["update_ms"]))
    print("H %3d: rollout %.4f ms  update %.4f ms" % res[-1])
    plan.close()
(h0, r0, _), (h1, r1, _) = res[1], res[-1]
step = (r1 - r0) / (h1 - h0)
print("per step %.3f us, fixed %.1f us" % (1e3 * step, 1e3 * (r0 - step * h0)))
