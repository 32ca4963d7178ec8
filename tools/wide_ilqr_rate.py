"""iLQR on wide linear models (csrc/ilqr_wide.hpp): ARX-shaped random stable models of 66 .. 128 states,
256 problems through 256 slots (ampc_ilqr_solve_queue), horizon 25; per-iteration kernel times.
python tools/wide_ilqr_rate.py"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from autompc_amd import _lib                                          # noqa: E402

for ns, nu, no in ((66, 6, 18), (91, 6, 18), (128, 8, 18)):
    rng = np.random.default_rng(ns)
    A = 0.92 * np.linalg.qr(rng.normal(size=(ns, ns)))[0]
    B = rng.normal(scale=0.3, size=(ns, nu))
    h = _lib.Handle(0, "f64")
    h.set_linear(A, B)
    h.set_quad_costs(np.eye(no), 0.1 * np.eye(nu), np.eye(no), np.zeros(no))
    h.set_ctrl_bounds(np.full(nu, -0.3), np.full(nu, 0.3))
    P, H = 256, 25
    x0 = rng.uniform(-0.5, 0.5, size=(P, ns))
    plan = _lib.IlqrPlan(h, 256, H, 0.05, clip_to_bounds=True)
    plan.solve_queue(x0[:32], max_iter=50, gains=False, trajectories=False)
    t0 = time.perf_counter()
    q = plan.solve_queue(x0, max_iter=50, gains=False, trajectories=False)
    dt = time.perf_counter() - t0
    plan.set_timing(True)
    plan.solve_queue(x0, max_iter=50, gains=False, trajectories=False)
    tm = plan.timing()
    # (linear-quadratic problems converge in two iterations: the per-launch averages below include the launches
    #  in which most slots were already idle -- they are launch times, not a roofline figure)
    print("ns %3d nu %d: %d problems H %d: %.1f ms = %.0f solves/s, mean iters %.1f, converged %.2f; per launch "
          "(averaged over %d): sweep %.3f ms, line search %.3f ms"
          % (ns, nu, P, H, 1e3 * dt, P / dt, q["iters"].mean(), q["converged"].mean(), tm["launches"],
             tm["riccati_ms"], tm["iter_ms"]))
    plan.close()
    h.close()
