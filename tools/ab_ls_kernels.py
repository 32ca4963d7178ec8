"""Lock-step batches (ampc_ilqr_solve, 4 x 256 HalfCheetah problems) with the line-search kernel forced
(AMPC_LS4_RB=1: four-row passes, =3: one twelve-row pass) or chosen per poll (unset / 0).  UNBOUNDED=1: the
never-converging problem set.  python tools/ab_ls_kernels.py"""
import sys, time, os, numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from autompc_amd import _lib
from autompc_amd.synthetic import make_workload
system, task, model, spec = make_workload("c3", precision="f64", device=0)
nx, nu = spec["nx"], spec["nu"]
Q, R, F = task.get_cost().get_cost_matrices()
h = _lib.Handle(0, "f64"); model.stage_into(h)
h.set_quad_costs(Q, R, F, task.get_cost().get_goal())
bounded = os.environ.get("UNBOUNDED") is None
if bounded: h.set_ctrl_bounds(np.full(nu, -0.25), np.full(nu, 0.25))
x0 = np.random.default_rng(0).uniform(-0.1, 0.1, size=(1024, nx))
B = 256
plan = _lib.IlqrPlan(h, B, 50, system.dt, clip_to_bounds=bounded)
for rep in range(4):
    t0 = time.perf_counter()
    for lo in range(0, 1024, B):
        o = plan.solve(x0[lo:lo + B], np.zeros((B, 50, nu)), max_iter=50)
    dt = time.perf_counter() - t0
    print("AMPC_LS4_RB=%s bounded=%s rep %d: %.1f ms (%.0f solves/s) rows %d" % (os.environ.get("AMPC_LS4_RB", "auto"), bounded, rep, 1e3 * dt, 1024 / dt, plan.stats()["candidate_rows"]))
