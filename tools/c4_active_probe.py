"""Experiment: what an iLQR iteration launch costs as a function of the slots still active (lock-step batch of B
problems of the converging c4 set): if it were proportional, admitting all problems at once would have no drain.
    python tools/c4_active_probe.py [B] [queue]        (on the GPU box)"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from autompc_amd import _lib                                   # noqa: E402
from autompc_amd.synthetic import make_workload                # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
QUEUE = len(sys.argv) > 2 and sys.argv[2] == "queue"      # through ampc_ilqr_solve_queue (slots with work first)
system, task, model, spec = make_workload("c3", precision="f64", device=0)
nx, nu = spec["nx"], spec["nu"]
Q, R, F = task.get_cost().get_cost_matrices()
h = _lib.Handle(0, "f64")
model.stage_into(h)
h.set_quad_costs(Q, R, F, task.get_cost().get_goal())
h.set_ctrl_bounds(np.full(nu, -0.25), np.full(nu, 0.25))
x0 = np.random.default_rng(0).uniform(-0.1, 0.1, size=(B, nx))
ug = np.zeros((B, 50, nu))
plan = _lib.IlqrPlan(h, B, 50, system.dt, clip_to_bounds=True)
full = plan.solve(x0, ug, 50)
it = full["iters"].astype(int)
prev_t, prev_k, prev = 0.0, 0, None
for mi in (1, 2, 4, 6, 8, 10, 13, 16, 20, 25, 30, 40, 50):
    plan.set_timing(True)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        if QUEUE:
            plan.solve_queue(x0, max_iter=mi, gains=False, trajectories=False)
        else:
            plan.solve(x0, ug, mi)
        ts.append(time.perf_counter() - t0)
    dt = 1e3 * min(ts)
    t = plan.timing()
    tot = {k: t[k] * t["launches"] for k in ("riccati_ms", "iter_ms", "forward_ms", "jacobian_ms")}
    act = [(it >= k).sum() for k in range(prev_k + 1, mi + 1)]
    line = "iterations %2d..%2d  active %4d..%4d  %.3f ms per iteration" % (prev_k + 1, mi, act[0], act[-1], (dt - prev_t) / (mi - prev_k))
    if prev is not None:
        n = t["launches"] - prev[1]
        line += "   kernels per launch: " + "  ".join("%s %.3f" % (k[:-3], (tot[k] - prev[0][k]) / max(n, 1)) for k in tot)
    print(line)
    prev_t, prev_k, prev = dt, mi, (tot, t["launches"])
