"""Per-wave timeline of ONE time step of the twelve-row line search (ilqr_lsw.hpp): every wave of workgroup 7
keeps s_memtime marks in registers (needs the AMPC_X_PHASETIME build: variants/lib_phasetime.so).
  python tools/phasetime_lsw.py [B]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AMPC_LIB"] = os.path.join(ROOT, "variants", "lib_phasetime.so")
os.environ["AMPC_LS4_PAR"] = "0"
os.environ["AMPC_LS4_RB"] = "3"
from autompc_amd import _lib                                   # noqa: E402
from autompc_amd.synthetic import make_workload                # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
system, task, model, spec = make_workload("c3", precision="f64")
h = _lib.Handle(0, "f64")
model.stage_into(h)
Q, R, F = task.get_cost().get_cost_matrices()
h.set_quad_costs(Q, R, F, task.get_cost().get_goal())
nx, nu = spec["nx"], spec["nu"]
h.set_ctrl_bounds(np.full(nu, -0.25), np.full(nu, 0.25))
plan = _lib.IlqrPlan(h, B, 50, system.dt, clip_to_bounds=True)
x0 = np.random.default_rng(0).uniform(-0.1, 0.1, size=(B, nx))
plan.solve(x0, np.zeros((B, 50, nu)), 5)
h.synchronize()
marks = (ctypes.c_longlong * 128)()
lib = _lib.load()
lib.ampc_x_phase_marks_ilqr.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
lib.ampc_x_phase_marks_ilqr(marks)
mm = np.array(marks[64:128], dtype=np.int64).reshape(4, 16)
m = mm[:, :8]
names = ["state update + control law", "barrier", "layer 0", "barrier", "hidden layer + barrier", "output layer", "barrier"]
t0 = m[:, 0].min()
print("B = %d; marks relative to the earliest wave's start of the step (s_memtime ticks)" % B)
print("%-30s %s" % ("phase", "  ".join("wave %d" % w for w in range(4))))
for i, n in enumerate(names):
    print("%-30s %s" % (n, "  ".join("%6d" % (m[w, i + 1] - m[w, i]) for w in range(4))))
print("%-30s %s" % ("step (mark 0 -> mark 7)", "  ".join("%6d" % (m[w, 7] - m[w, 0]) for w in range(4))))
print("%-30s %s" % ("start offsets", "  ".join("%6d" % (m[w, 0] - t0) for w in range(4))))
print("%-30s %s" % ("entry -> time loop", "  ".join("%6d" % (mm[w, 9] - mm[w, 8]) for w in range(4))))
print("%-30s %s" % ("time loop (51 steps)", "  ".join("%6d" % (mm[w, 10] - mm[w, 9]) for w in range(4))))
print("%-30s %s" % ("objectives", "  ".join("%6d" % (mm[w, 11] - mm[w, 10]) for w in range(4))))
