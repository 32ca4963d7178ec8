"""Experiment: per-phase shader-clock breakdown of one backward Riccati step of the f64 iLQR
kernel (needs the AMPC_X_PHASETIME build: tools/variants.sh -> variants/lib_phasetime.so)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AMPC_LIB"] = os.path.join(ROOT, "variants", "lib_phasetime.so")
from autompc_amd import _lib
from autompc_amd.synthetic import make_workload
system, task, model, spec = make_workload("c3", precision="f64")
h = _lib.Handle(0, "f64")
model.stage_into(h)
Q, R, F = task.get_cost().get_cost_matrices()
h.set_quad_costs(Q, R, F, task.get_cost().get_goal())
B, H, nx, nu = 64, 50, spec["nx"], spec["nu"]
plan = _lib.IlqrPlan(h, B, H, system.dt)
x0 = np.tile(task.get_init_obs(), (B, 1)) + np.random.default_rng(0).uniform(-0.01, 0.01, size=(B, nx))
plan.solve(x0, np.zeros((B, H, nu)), 5)
h.synchronize()
marks = (ctypes.c_longlong * 128)()
lib = _lib.load()
lib.ampc_x_phase_marks_ilqr.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
lib.ampc_x_phase_marks_ilqr(marks)
m = np.array(marks[:56], dtype=np.int64)
if os.environ.get("AMPC_RICCATI") == "0":
    names = {21: "fetch issue + VJ = V J (+bar)", 22: "Qt, qt (+bar)", 23: "Gauss-Jordan (wave 0)",
             24: "barrier", 25: "Wk, wq, sums (+bar)", 26: "V, v update, commit (+bar)"}
    seq = [20, 21, 22, 23, 24, 25, 26]
else:
    names = {10: "A: VJ = V J (wave 0's tile)", 21: "barrier", 11: "B: Qt = Ct + J'[VJ|v]", 22: "barrier",
             23: "C: per-lane LU solve, K, Z", 24: "barrier", 25: "D: V, v update", 26: "barrier"}
    seq = [20, 10, 21, 11, 22, 23, 24, 25, 26]
print("Riccati step, cycles:", m[26] - m[20], "(each line includes the mark's own s_memtime wait)")
for a, b in zip(seq[:-1], seq[1:]):
    print("  %-32s %6d" % (names[b], m[b] - m[a]))
print("problem 7, last iteration: Riccati sweep %d cycles (ilqr_riccati_kernel), line-search rollout %d cycles (ilqr_iter_kernel)" % (m[33] - m[30], m[32] - m[31]))
if os.environ.get("AMPC_LS4") == "0":
    ls = {41: "fetch issue + controls + lss stores", 42: "barrier", 43: "objective", 44: "network (net.run)",
          45: "reduce + state update", 46: "commit + barrier"}
    seq = [40, 41, 42, 43, 44, 45, 46]
else:
    ls = {41: "state update + controls + objective", 42: "barrier", 44: "layer 0", 45: "barrier",
          46: "hidden layer(s) incl. barrier", 47: "output layer partials", 48: "barrier"}
    seq = [40, 41, 42, 44, 45, 46, 47, 48]
m = np.array(marks[:56], dtype=np.int64)
print("line-search step, cycles:", m[seq[-1]] - m[40], "(each line includes the mark's own s_memtime wait)")
for a, b in zip(seq[:-1], seq[1:]):
    print("  %-36s %6d" % (ls[b], m[b] - m[a]))
