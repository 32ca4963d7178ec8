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
marks = (ctypes.c_longlong * 128)()
lib = _lib.load()
lib.ampc_x_phase_marks.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
lib.ampc_x_phase_marks(marks)
m = np.array(marks[:16], dtype=np.int64)
# mark ids in program order (mlp_tile.hpp / mppi_kernels.hpp AMPC_MARK)
order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 13, 9, 12, 10, 11]
names = {1: "dense cost", 2: "layer0 mma", 3: "epi0 (+bar)", 4: "hidden mma (+bar inside)", 5: "prefetch (+bar)",
         6: "epi1", 7: "out mma", 8: "prefetch0", 13: "partials write", 9: "barrier", 12: "reduce + state update",
         10: "actions", 11: "barrier"}
print("precision", prec, "batch", batch, "total cycles/step", m[11] - m[0])
for a, b in zip(order[:-1], order[1:]):
    print("  %-26s %6d" % (names[b], m[b] - m[a]))
