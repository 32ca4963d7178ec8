"""Experiment: timeline of one rollout time step for all 8 waves of one workgroup (needs the
AMPC_X_WAVETIME build: python tools/ab_variants.py wavetime:-DAMPC_X_WAVETIME).  Marks are kept in
registers and written once at kernel end, so the only perturbation is the s_memtime itself."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AMPC_LIB"] = os.path.join(ROOT, "variants", sys.argv[1] if len(sys.argv) > 1 else "lib_wavetime.so")
PREC = sys.argv[2] if len(sys.argv) > 2 else "f64"          # python tools/wavetime.py lib_wavetime.so f32
from autompc_amd import _lib
from autompc_amd.synthetic import make_workload
system, task, model, spec = make_workload("c3", precision=PREC)
h = _lib.Handle(0, PREC)
model.stage_into(h)
Q, R, F = task.get_cost().get_cost_matrices()
h.set_quad_costs(Q, R, F, task.get_cost().get_goal())
b = task.get_ctrl_bounds(); h.set_ctrl_bounds(b[:, 0], b[:, 1])
N, H, nu = spec["num_path"], spec["horizon"], spec["nu"]
plan = _lib.MppiPlan(h, [N], [H], [1.0], [1.0])
plan.upload(task.get_init_obs(), np.zeros(H * nu))
plan.set_outputs(keep_eps_out=False)
for i in range(30):
    plan.generate_eps(0, i)
    plan.solve()
h.synchronize()
marks = (ctypes.c_longlong * 128)()
lib = _lib.load()
fn = lib.ampc_x_wave_marks if PREC == "f64" else lib.ampc_x_wave_marks_f32      # (one copy of the marks per translation unit)
fn.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
fn(marks)
m = np.array(marks[:], dtype=np.int64).reshape(8, 16)
order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 13, 9, 12, 10, 11]
names = {0: "step start", 1: "dense cost done", 2: "L0 mma issued", 3: "epi0 done", 4: "L1 mma issued (+bar inside)",
         5: "prefetch/bar", 6: "epi1 done", 7: "out mma issued", 8: "prefetch0+side done", 13: "partials written",
         9: "barrier B1 passed", 12: "reduce+update done", 10: "(actions)", 11: "barrier B2 passed"}
t0 = m[:, 0].min()
print("cycles relative to the earliest wave's step start; one column per wave (w0..w7; w and w+4 share a SIMD)")
for k in order:
    print("%-30s" % names[k], " ".join("%6d" % (m[w, k] - t0) for w in range(8)))
print("step length (wave 0, start -> B2 passed):", m[0, 11] - m[0, 0])
