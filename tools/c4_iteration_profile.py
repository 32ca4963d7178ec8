import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from autompc_amd import _lib
from autompc_amd.synthetic import make_workload
system, task, model, spec = make_workload("c3")
nx, nu, B = spec["nx"], spec["nu"], 256
Q, Rm, F = task.get_cost().get_cost_matrices()
for bounded in (False, True):
    h = _lib.Handle(0, "f64")
    model.stage_into(h)
    h.set_quad_costs(Q, Rm, F, task.get_cost().get_goal())
    if bounded:
        h.set_ctrl_bounds(np.full(nu, -0.25), np.full(nu, 0.25))
    plan = _lib.IlqrPlan(h, B, 50, system.dt, clip_to_bounds=bounded)
    x0 = np.random.default_rng(0).uniform(-0.1, 0.1, size=(B, nx))
    ug = np.zeros((B, 50, nu))
    plan.solve(x0, ug, 50)
    for mi in (5, 10, 20, 30, 50):
        plan.set_timing(True)
        t0 = time.perf_counter(); out = plan.solve(x0, ug, mi); dt = time.perf_counter() - t0
        t = plan.timing()
        print("bounded=%d max_iter=%2d  %.2f ms  active at end %3d  mean it %.1f  per-iteration: sweep %.3f ls %.3f fwd %.3f jac %.3f (n=%d)"
              % (bounded, mi, 1e3 * dt, int(B - out["converged"].sum()) if mi == 50 else -1, out["iters"].mean(),
                 t["riccati_ms"], t["iter_ms"], t["forward_ms"], t["jacobian_ms"], t["launches"]))
    plan.close(); h.close()
