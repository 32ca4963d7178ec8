"""Kernels specialised at run time for the staged model's shape (csrc/jit_host.hpp, shapes.hpp): the
reference's MLP configuration space is 1-4 hidden layers of 16-256 units (mlp.py:113-122); shapes
outside the build-time registry get their StaticShape kernels from a plugin compiled by hipcc on
first use.  Same arithmetic in the same order: every result must be bit-identical to the
run-time-shape kernels.  Needs MI355X (and hipcc, which the image has)."""
import numpy as np
import pytest

from helpers import rel_err
from oracle import mlp as omlp

pytestmark = pytest.mark.gpu

SHAPES = [
    # nx, nu, hidden, activation  (none of these is in csrc/shapes.hpp)
    (11, 5, [256, 256], "relu"),
    (12, 3, [192, 160, 192], "tanh"),
    (5, 2, [40], "relu"),
]


def _handle(nx, nu, hidden, act, precision, seed=3):
    from autompc_amd import _lib
    p = omlp.random_params(nx, nu, hidden, act, seed=seed)
    h = _lib.Handle(0, precision)
    h.set_mlp(nx, nu, p["weights"], p["biases"], act, p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"])
    h.set_quad_costs(np.eye(nx), 0.05 * np.eye(nu), 2 * np.eye(nx), np.zeros(nx))
    h.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
    return h


def _mppi(h, nx, nu, tile_rows):
    from autompc_amd import _lib
    N, H = 512, 12
    plan = _lib.MppiPlan(h, [N], [H], [0.5], [0.7])
    if tile_rows:
        plan.set_geometry(tile_rows, 0)
    rng = np.random.default_rng(5)
    plan.upload(rng.uniform(-0.2, 0.2, size=(1, nx)), rng.normal(scale=0.3, size=H * nu),
                rng.normal(scale=0.7, size=N * H * nu))
    plan.solve()
    out = plan.download(costs=True, eps_out=True)
    kind = plan.kernel_kind()
    plan.close()
    return kind, out


def _ilqr(h, nx, nu):
    from autompc_amd import _lib
    B, H = 3, 20
    plan = _lib.IlqrPlan(h, B, H, 0.05)
    x0 = np.random.default_rng(2).uniform(-0.3, 0.3, size=(B, nx))
    out = plan.solve(x0, np.zeros((B, H, nu)), max_iter=15)
    kind = plan.kernel_kind()
    plan.close()
    return kind, out


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("shape", SHAPES)
def test_jit_kernels_equal_the_runtime_shape_kernels(shape, precision, monkeypatch, tmp_path):
    nx, nu, hidden, act = shape
    if shape == SHAPES[2] and precision == "f64":
        # one case goes through a fresh cache directory: the build itself is exercised, not a cached
        # plugin of an earlier run
        monkeypatch.setenv("AMPC_JIT_CACHE", str(tmp_path))
    monkeypatch.setenv("AMPC_JIT", "0")
    h = _handle(nx, nu, hidden, act, precision)
    assert h.jit_status()[0] == 0
    ref_m = {tr: _mppi(h, nx, nu, tr) for tr in (16, 32)}
    ref_i = _ilqr(h, nx, nu)
    assert all(k == 0 for k, _ in ref_m.values()) and ref_i[0] == 0
    h.close()

    monkeypatch.setenv("AMPC_JIT", "1")
    h = _handle(nx, nu, hidden, act, precision)
    st, msg = h.jit_status()
    assert st in (1, 2), msg
    # a plan created while the plugin is still compiling runs the run-time-shape kernels
    early_kind, early = _mppi(h, nx, nu, 16)
    assert early_kind in (0, 2)
    h.jit_wait()
    st, msg = h.jit_status()
    assert st == 2 and msg.endswith(".so"), msg
    for tr in (16, 32):
        kind, got = _mppi(h, nx, nu, tr)
        assert kind == 2, "plan did not pick the run-time compiled kernels"
        for a, b in zip(got, ref_m[tr][1]):
            np.testing.assert_array_equal(a, b)
    for a, b in zip(early, ref_m[16][1]):
        np.testing.assert_array_equal(a, b)
    kind, got = _ilqr(h, nx, nu)
    assert kind == 2
    ref = ref_i[1]
    if precision == "f64":
        assert np.array_equal(got["iters"], ref["iters"]) and np.array_equal(got["converged"], ref["converged"])
        for key in ("states", "ctrls", "Ks", "ks", "objective"):
            np.testing.assert_array_equal(got[key], ref[key])
    else:
        # f32 iLQR (outside the parity mode, control/ilqr.py): with compile-time extents hipcc forms
        # packed f32 multiplies / adds in the sweep's scalar loops where the run-time-shape build
        # emits fused multiply-adds -- rounding-level differences (measured 5e-6 relative)
        assert rel_err(got["states"], ref["states"]) < 1e-4 and rel_err(got["ctrls"], ref["ctrls"]) < 1e-3
    h.close()


def test_registered_shape_needs_no_plugin():
    from autompc_amd import _lib
    h = _handle(17, 6, [256, 256], "relu", "f64")        # shapes.hpp entry 0
    assert h.jit_status()[0] == 0
    h.jit_wait()
    plan = _lib.MppiPlan(h, [256], [10], [1.0], [1.0])
    assert plan.kernel_kind() == 1
    plan.close()
    h.close()


def test_small_linear_models_get_a_plugin_with_the_same_bits(monkeypatch):
    """A linear model of up to 32 states is staged through the MLP tile (one identity layer): it gets
    shape-specialised kernels like any MLP (round 5: the 20-state ARX bench line 13.9 k -> 24.4 k solves/s),
    with the bits of the run-time-shape kernels."""
    from autompc_amd import _lib
    rng = np.random.default_rng(5)
    nx, nu, N, H = 11, 2, 300, 12
    A = 0.9 * np.eye(nx) + 0.02 * rng.normal(size=(nx, nx))
    B = 0.1 * rng.normal(size=(nx, nu))
    x0 = rng.uniform(-0.5, 0.5, size=(1, nx))
    out = {}
    for jit in ("0", "1"):
        monkeypatch.setenv("AMPC_JIT", jit)
        h = _lib.Handle(0, "f64")
        h.set_linear(A, B)
        h.set_quad_costs(np.eye(nx), 0.1 * np.eye(nu), 2.0 * np.eye(nx), np.zeros(nx))
        h.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
        if jit == "1":
            assert h.jit_status()[0] in (1, 2)
            h.jit_wait()
            assert h.jit_status()[0] == 2, h.jit_status()
        plan = _lib.MppiPlan(h, [N], [H], [0.7], [0.9])
        plan.upload(x0, np.zeros(H * nu), None)
        plan.generate_eps(3, 1)
        plan.solve()
        a, u, c, _ = plan.download(costs=True)
        ip = _lib.IlqrPlan(h, 3, H, 0.05)
        o = ip.solve(np.repeat(x0, 3, axis=0) * np.array([[1.0], [0.5], [-1.0]]), np.zeros((3, H, nu)), 8)
        out[jit] = (a, u, c, o["ctrls"], plan.kernel_kind(), ip.kernel_kind())
        ip.close()
        plan.close()
        h.close()
    assert out["1"][4] == 2 and out["1"][5] == 2
    for k in range(4):
        assert np.array_equal(out["0"][k], out["1"][k])


def test_controllers_switch_to_the_compiled_kernels_mid_run(monkeypatch, tmp_path):
    """A drop-in controller keeps its plan for its lifetime; when the shape plugin finishes
    compiling after the first run() the controller moves its warm start into a new plan on the
    specialised kernels.  The sequence of controls is the one the run-time-shape kernels give."""
    from autompc_amd import MLP, MPPI, IterativeLQR, QuadCost, Task
    from helpers import make_system
    nx, nu, hidden = 7, 2, [96, 80]
    system = make_system(nx, nu)
    p = omlp.random_params(nx, nu, hidden, "tanh", seed=12)

    def model():
        m = MLP(system, n_hidden_layers=2, hidden_size_1=96, hidden_size_2=80, nonlintype="tanh")
        m.weights, m.biases = [w.copy() for w in p["weights"]], [b.copy() for b in p["biases"]]
        m.xu_means, m.xu_std, m.dy_means, m.dy_std = p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"]
        return m
    task = Task(system)
    task.set_cost(QuadCost(system, np.eye(nx), 0.1 * np.eye(nu), np.eye(nx)))
    task.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
    obs0 = np.random.default_rng(0).uniform(-0.2, 0.2, size=nx)

    def drive(wait_after_first):
        np.random.seed(3)
        ctl = MPPI(system, task, model(), horizon=8, num_path=128, sigma=0.5, lmda=0.8)
        ilq = IterativeLQR(system, task, model(), 10)
        cs, obs, us, kinds = np.concatenate([obs0, np.zeros(nu)]), obs0.copy(), [], []
        for k in range(4):
            u, cs = ctl.run(cs, obs)
            ui, _ = ilq.run(np.concatenate([obs, np.zeros(nu)]), obs)
            us.append(np.concatenate([u, ui]))
            kinds.append((ctl._plan.kernel_kind(), ilq._plan.kernel_kind()))
            obs = obs + 0.05 * np.tanh(u).sum() * np.ones(nx)
            if k == 0 and wait_after_first:
                ctl._handle.jit_wait()
                ilq._handle.jit_wait()
        return np.array(us), kinds
    monkeypatch.delenv("AMPC_QUAD", raising=False)
    monkeypatch.setenv("AMPC_JIT", "0")
    ref, kinds0 = drive(False)
    assert all(k == (3, 0) for k in kinds0)       # (128 samples: the four-row rollout kernel, run-time shape)
    monkeypatch.setenv("AMPC_JIT", "1")
    monkeypatch.setenv("AMPC_JIT_CACHE", str(tmp_path))       # nothing cached: the build runs now
    got, kinds1 = drive(True)
    assert kinds1[-1] == (2, 2), kinds1
    np.testing.assert_array_equal(got, ref)


def test_failed_build_falls_back_to_the_runtime_shape_kernels(monkeypatch):
    """A plugin that cannot be built (here: a cache directory that cannot be created) must never
    break a solve: the run-time-shape kernels keep serving, the status says what happened."""
    from autompc_amd import _lib
    monkeypatch.setenv("AMPC_JIT", "1")
    monkeypatch.setenv("AMPC_JIT_CACHE", "/proc/no_such_dir/ampc")
    nx, nu = 6, 2
    h = _handle(nx, nu, [72, 72], "relu", "f64", seed=4)
    with pytest.raises(_lib.AmpcError):
        h.jit_wait()
    st, msg = h.jit_status()
    assert st == -1 and msg
    kind, out = _mppi(h, nx, nu, 16)
    assert kind == 0 and np.all(np.isfinite(out[1]))
    h.close()


def test_single_hidden_layer_specialised_rollout_next_to_other_kernels(monkeypatch, tmp_path):
    """Regression (tools/fuzz_gpu.py seed 31 case 155): with ONE hidden layer nothing separated the
    first layer's reads of [x | u] from the next step's controls being written into it; the
    specialised eight-wave kernel, running next to the noise generator's side stream, then rolled
    out the samples of its faster waves with the wrong controls (cost errors of 1e-2, different
    from run to run).  The drop-in controller in its default noise mode against the oracle."""
    from autompc_amd import MLP, MPPI, QuadCost, Task
    from helpers import make_system
    from oracle.costs import QuadCostOracle
    from oracle.mlp import MLPOracle
    from oracle.mppi import MPPIOracle
    monkeypatch.setenv("AMPC_JIT", "1")
    monkeypatch.setenv("AMPC_JIT_CACHE", str(tmp_path))
    nx, nu, N, H = 16, 2, 64, 15
    system = make_system(nx, nu)
    rng = np.random.default_rng(155)
    p = omlp.random_params(nx, nu, [100], "relu", seed=31)
    A = rng.normal(size=(nx, nx))
    Q, R, F, goal = A @ A.T / nx + 0.1 * np.eye(nx), np.diag(rng.uniform(0.01, 0.1, nu)), np.eye(nx), rng.normal(size=nx) * 0.1
    task = Task(system)
    task.set_cost(QuadCost(system, Q, R, F, goal=goal))
    task.set_ctrl_bounds(np.full(nu, -0.8), np.full(nu, 1.1))
    obs = rng.uniform(-0.1, 0.1, size=nx)
    for mt in ("1", "2"):
        monkeypatch.setenv("AMPC_MT", mt)
        for rep in range(4):
            m = MLP(system, n_hidden_layers=1, hidden_size_1=100, nonlintype="relu")
            m.weights, m.biases = [w.copy() for w in p["weights"]], [b.copy() for b in p["biases"]]
            m.xu_means, m.xu_std, m.dy_means, m.dy_std = p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"]
            np.random.seed(7)
            orc = MPPIOracle(MLPOracle(system, p), QuadCostOracle(Q, R, F, goal), np.tile([-0.8, 1.1], (nu, 1)),
                             horizon=H, num_path=N, sigma=0.6, lmda=0.9)
            np.random.seed(7)
            ctl = MPPI(system, task, m, horizon=H, num_path=N, sigma=0.6, lmda=0.9)
            ctl._device()
            ctl._handle.jit_wait()
            assert ctl._device().kernel_kind() == 2 and ctl._device().info()["samples_per_wg"] == 16 * int(mt)
            cs = np.concatenate([obs, np.zeros(nu)])
            st = np.random.get_state()
            orc.run(cs, obs)
            np.random.set_state(st)
            ctl.run(cs, obs, return_details=True)
            assert rel_err(ctl.last_costs, orc.last_costs) < 1e-9, (mt, rep)


def test_long_lived_controller_switches_over_without_jit_wait(monkeypatch, tmp_path):
    """nobody calls ampc_jit_wait in production.  A controller that just keeps calling
    run() must move to the compiled kernels by itself once the background build has finished
    (ampc_jit_status reaps the build), with unchanged results, and the hipcc child must not stay a
    zombie."""
    import os
    import time
    from autompc_amd import MLP, MPPI, QuadCost, Task
    from helpers import make_system
    monkeypatch.delenv("AMPC_QUAD", raising=False)
    monkeypatch.setenv("AMPC_JIT", "1")
    monkeypatch.setenv("AMPC_JIT_CACHE", str(tmp_path))
    nx, nu = 11, 3                     # (a shape no other test of this module compiles: the table is per process)
    system = make_system(nx, nu)
    p = omlp.random_params(nx, nu, [80, 80], "tanh", seed=8)
    m = MLP(system, n_hidden_layers=2, hidden_size_1=80, hidden_size_2=80, nonlintype="tanh")
    m.weights, m.biases = [w.copy() for w in p["weights"]], [b.copy() for b in p["biases"]]
    m.xu_means, m.xu_std, m.dy_means, m.dy_std = p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"]
    task = Task(system)
    task.set_cost(QuadCost(system, np.eye(nx), 0.1 * np.eye(nu), np.eye(nx)))
    task.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
    np.random.seed(1)                  # (the warm start is drawn from numpy's global stream)
    ctl = MPPI(system, task, m, horizon=8, num_path=2048, noise="device")
    cs, obs = np.zeros(nx + nu), np.full(nx, 0.1)
    u0, _ = ctl.run(cs, obs)
    assert ctl._plan.kernel_kind() in (0, 3) and ctl._handle.jit_status()[0] == 1      # run-time shape; still building
    t0, kind = time.time(), 0
    while time.time() - t0 < 120:
        ctl.run(cs, obs)
        kind = ctl._plan.kernel_kind()
        if kind == 2:
            break
        time.sleep(0.05)
    assert kind == 2, "the controller never switched to the compiled kernels"
    # same numbers from the compiled kernels (fresh controller, same Philox stream)
    np.random.seed(1)
    ctl2 = MPPI(system, task, m, horizon=8, num_path=2048, noise="device")
    u2, _ = ctl2.run(cs, obs)
    assert ctl2._plan.kernel_kind() == 2
    np.testing.assert_array_equal(u2, u0)
    # no zombie child left behind
    me = os.getpid()
    zombies = []
    for pid in os.listdir("/proc"):
        if pid.isdigit():
            try:
                f = open("/proc/%s/stat" % pid).read().rsplit(")", 1)[1].split()
            except OSError:
                continue
            if f[0] == "Z" and int(f[1]) == me:
                zombies.append(pid)
    assert not zombies


def test_processes_wanting_the_same_plugin_share_one_build(tmp_path):
    """the ranks of a torch.distributed job stage the same model at the same time.
    They must share ONE hipcc build (lock directory in the cache) instead of racing eight compiles:
    three processes started together all end up with the plugin, and only one build log shows
    compiler invocations."""
    import glob
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from autompc_amd import _lib
from oracle import mlp as omlp
p = omlp.random_params(9, 2, [90, 90], "relu", seed=1)
h = _lib.Handle(0, "f64")
h.set_mlp(9, 2, p["weights"], p["biases"], "relu", p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"])
h.set_quad_costs(np.eye(9), np.eye(2), np.eye(9), np.zeros(9))
h.jit_wait()
print("STATUS", h.jit_status()[0])
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AMPC_JIT="1", AMPC_JIT_CACHE=str(tmp_path))
    procs = [subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for _ in range(3)]
    outs = [pr.communicate(timeout=300)[0].decode() for pr in procs]
    for o in outs:
        assert "STATUS 2" in o, o
    assert len(glob.glob(str(tmp_path / "shape_*.so"))) == 1
    assert not glob.glob(str(tmp_path / "*.lock")) and not glob.glob(str(tmp_path / "build_*"))
    # exactly one of the three scripts compiled; the others waited for its result
    built = [f for f in glob.glob(str(tmp_path / "shape_*.log")) if os.path.getsize(f) >= 0]
    assert len(built) == 3
