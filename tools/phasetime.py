"""Experiment: per-phase shader-clock breakdown of one rollout time step (needs the
AMPC_X_PHASETIME build: tools/variants.sh -> variants/lib_phasetime.so)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AMPC_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "variants", "lib_phasetime.so")
from autompc_amd import _lib
from autompc_amd.synthetic import make_workload
prec = sys.argv[1] if len(sys.argv) > 1 else "f64"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
system, task, model, spec = make_workload("c3", precision=prec)
h = _lib.Handle(0, prec)
model.stage_into(h)
Q, R, F = task.get_cost().get_cost_matrices()
h.set_quad_costs(Q, R, F, task.get_cost().get_goal())
b = task.get_ctrl_bounds(); h.set_ctrl_bounds(b[:, 0], b[:, 1])
N, H, nu, nx = spec["num_path"], spec["horizon"], spec["nu"], spec["nx"]
plan = _lib.MppiPlan(h, [N] * batch, [H] * batch, [1.0] * batch, [1.0] * batch)
plan.upload(np.tile(task.get_init_obs(), (batch, 1)), np.zeros(batch * H * nu))
plan.generate_eps(0, 0)
for _ in range(3):
    plan.solve()
h.synchronize()
marks = (ctypes.c_longlong * 64)()
lib = _lib.load()
lib.ampc_x_phase_marks.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
lib.ampc_x_phase_marks(marks)
m = np.array(marks[:12], dtype=np.int64)
names = ["cost", "layer0 mma", "epi0+bar", "hidden mma", "bar", "epi1+bar", "out mma", "pf0+bar", "partials+bar", "update+actions", "bar"]
print("precision", prec, "batch", batch, "total cycles/step", m[11] - m[0])
for i, nme in enumerate(names):
    print("  %-16s %6d" % (nme, m[i + 1] - m[i]))
