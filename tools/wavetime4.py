"""Experiment: timeline of one time step of the FOUR-row rollout kernel for the four waves of one
workgroup (AMPC_X_WAVETIME build: python tools/ab_variants.py wavetime:-DAMPC_X_WAVETIME).
Usage: python tools/wavetime4.py [lib] [rows]   rows = 4 (default) or 16 (marks of mppi_rollout_kernel)"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AMPC_LIB"] = os.path.join(ROOT, "variants", sys.argv[1] if len(sys.argv) > 1 else "lib_wavetime.so")
os.environ["AMPC_JIT"] = "0"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4
from autompc_amd import _lib
from autompc_amd.synthetic import make_workload
system, task, model, spec = make_workload("c2", precision="f64")
h = _lib.Handle(0, "f64")
model.stage_into(h)
Q, R, F = task.get_cost().get_cost_matrices()
h.set_quad_costs(Q, R, F, task.get_cost().get_goal())
b = task.get_ctrl_bounds(); h.set_ctrl_bounds(b[:, 0], b[:, 1])
N, H, nu = spec["num_path"], spec["horizon"], spec["nu"]
plan = _lib.MppiPlan(h, [N], [H], [1.0], [1.0])
plan.set_geometry(rows, 0)
plan.upload(task.get_init_obs(), np.zeros(H * nu))
plan.set_outputs(keep_eps_out=False)
for i in range(30):
    plan.generate_eps(0, i)
    plan.solve()
h.synchronize()
marks = (ctypes.c_longlong * 128)()
lib = _lib.load()
lib.ampc_x_wave_marks.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
lib.ampc_x_wave_marks(marks)
m = np.array(marks[:], dtype=np.int64).reshape(8, 16)
if rows == 4:
    names = ["step start", "barrier A passed", "L0 done (act written)", "barrier passed", "L1 done", "barrier passed",
             "out partials written", "next actions done", "barrier passed", "state updated"]
    t0 = m[:4, 0].min()
    for k, nm in enumerate(names):
        print("%-26s" % nm, " ".join("%6d" % (m[w, k] - t0) for w in range(4)))
else:
    t0 = m[:4, 0].min()
    for k in [0, 1, 2, 3, 4, 5, 6, 7, 8, 13, 9, 12, 10, 11]:
        print("mark %2d" % k, " ".join("%6d" % (m[w, k] - t0) for w in range(4)))
