"""c4 (HalfCheetah iLQR, H = 50) on the CONVERGING problem set: the batch solve (a batch lasts as long as
its slowest problem) against continuous batching (ampc_ilqr_solve_queue).  python tools/c4_queue_rate.py [P] [B ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from autompc_amd import _lib                                   # noqa: E402
from autompc_amd.synthetic import make_workload                # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
Bs = [int(a) for a in sys.argv[2:]] or [256]
system, task, model, spec = make_workload("c3", precision="f64", device=0)
nx, nu = spec["nx"], spec["nu"]
Q, R, F = task.get_cost().get_cost_matrices()
h = _lib.Handle(0, "f64")
model.stage_into(h)
h.set_quad_costs(Q, R, F, task.get_cost().get_goal())
h.set_ctrl_bounds(np.full(nu, -0.25), np.full(nu, 0.25))
rng = np.random.default_rng(0)
x0 = rng.uniform(-0.1, 0.1, size=(P, nx))
for B in Bs:
    plan = _lib.IlqrPlan(h, B, 50, system.dt, clip_to_bounds=True)
    plan.solve(x0[:B], np.zeros((B, 50, nu)), max_iter=50)
    t0 = time.perf_counter()
    its = []
    for lo in range(0, P, B):
        o = plan.solve(x0[lo:lo + B] if lo + B <= P else x0[-B:], np.zeros((B, 50, nu)), max_iter=50)
        its.append(o["iters"])
    tb = time.perf_counter() - t0
    plan.solve_queue(x0[:2 * B], max_iter=50, gains=False, trajectories=False)
    for kw in ({}, {"gains": False, "trajectories": False}):
        t0 = time.perf_counter()
        q = plan.solve_queue(x0, max_iter=50, **kw)
        tq = time.perf_counter() - t0
        st = plan.stats()
        print("B %4d  P %d  batch: %.1f ms (%.0f solves/s)   queue%s: %.1f ms (%.0f solves/s), %d iterations launched, "
              "mean iters %.1f, converged %.2f" % (B, P, 1e3 * tb, P / tb, " (scalars only)" if kw else "", 1e3 * tq, P / tq,
                                                   st["iterations"], q["iters"].mean(), q["converged"].mean()))
    plan.set_timing(True)
    plan.solve_queue(x0, max_iter=50, gains=False, trajectories=False)
    print("   per-iteration kernel ms:", plan.timing())
    plan.close()
