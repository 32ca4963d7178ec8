"""HIP MPPI solve vs the reference's golden vectors (nu = 1) and the oracle (nu > 1, full
BASELINE sizes).  Goes through Controller.run() -> ctypes -> C ABI -> HIP kernels."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden
from helpers import check_weights, cost_from_golden, golden_params, hip_cost_from_golden, make_system, rel_err
from oracle import mlp as omlp
from oracle.costs import QuadCostOracle
from oracle.mlp import MLPOracle
from oracle.mppi import MPPIOracle

pytestmark = pytest.mark.gpu


def _names():
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "mppi_*.npz")))


def _hip_stack(p, nx, nu, Q, R, F, goal, bounds, precision="f64", **mppi_kw):
    from autompc_amd import MLP, MPPI, QuadCost, Task
    system = make_system(nx, nu)
    m = MLP(system, n_hidden_layers=len(p["weights"]) - 1, nonlintype=p["activation"],
            precision=precision,
            **{"hidden_size_%d" % (i + 1): w.shape[0] for i, w in enumerate(p["weights"][:-1])})
    m.weights, m.biases = [w.copy() for w in p["weights"]], [b.copy() for b in p["biases"]]
    m.xu_means, m.xu_std, m.dy_means, m.dy_std = p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"]
    task = Task(system)
    task.set_cost(Q(system) if callable(Q) else QuadCost(system, Q, R, F, goal=goal))   # Q: or a cost builder
    task.set_ctrl_bounds(np.full(nu, bounds[0]), np.full(nu, bounds[1]))
    return system, m, task


@pytest.mark.parametrize("name", _names())
def test_mppi_matches_reference_golden(name):
    from autompc_amd import MPPI
    g = golden(name)
    nx = int(g["nx"])
    p = golden_params(nx, 1, g["hidden"], g["activation"], g["mlp_seed"], bool(g["plain_norm"]))
    check_weights(p, g)
    # (mppi_sumcost_*: sums of quadratic terms with different goals, mppi.py:73-82 over sum_cost.py:49-54)
    system, model, task = _hip_stack(p, nx, 1, lambda sy: hip_cost_from_golden(sy, g), None, None, None, g["bounds"])
    np.random.seed(int(g["np_seed"]))
    ctl = MPPI(system, task, model, horizon=int(g["H"]), num_path=int(g["N"]),
               sigma=float(g["sigma"]), lmda=float(g["lmda"]))
    np.testing.assert_array_equal(ctl.act_sequence, g["act0"])
    obs = np.random.default_rng(int(g["np_seed"]) + 99).uniform(-0.1, 0.1, size=nx)
    constate = np.concatenate([obs, np.zeros(1)])
    ref_model = MLPOracle(system, p)
    for r in range(int(g["n_runs"])):
        if r == 3:
            ctl.reset()
            np.testing.assert_array_equal(ctl.act_sequence, g["act_reset"])
        u, constate = ctl.run(constate, obs, return_details=True)
        assert rel_err(ctl.last_costs, g["costs_%d" % r]) < 1e-9
        assert rel_err(ctl.last_eps[:, ::16, :], g["eps_sub_%d" % r]) < 1e-12
        assert rel_err(ctl.act_sequence, g["act_%d" % r]) < 1e-8
        assert rel_err(u, g["u_%d" % r]) < 1e-8
        assert rel_err(constate, g["newstate_%d" % r]) < 1e-8
        obs = ref_model.pred(obs, g["u_%d" % r])


def _oracle_vs_hip(nx, nu, hidden, act, N, H, sigma, lmda, bounds, precision, tol, seed=0,
                   per_particle=False, runs=2):
    from autompc_amd import MPPI
    p = omlp.random_params(nx, nu, hidden, act, seed=seed + 5)
    rng = np.random.default_rng(seed)
    Q = np.diag(rng.uniform(0.5, 2.0, size=nx))
    R = np.diag(rng.uniform(0.01, 0.1, size=nu))
    F = np.diag(rng.uniform(0.5, 2.0, size=nx))
    goal = rng.normal(scale=0.1, size=nx)
    system, model, task = _hip_stack(p, nx, nu, Q, R, F, goal, bounds, precision)
    omodel = MLPOracle(system, p)
    bnd = np.tile(np.array(bounds), (nu, 1))
    np.random.seed(seed)
    orc = MPPIOracle(omodel, QuadCostOracle(Q, R, F, goal), bnd, horizon=H, num_path=N, sigma=sigma,
                     lmda=lmda, per_particle_terminal=per_particle)
    np.random.seed(seed)
    ctl = MPPI(system, task, model, horizon=H, num_path=N, sigma=sigma, lmda=lmda,
               per_particle_terminal=per_particle)
    obs = rng.uniform(-0.1, 0.1, size=nx)
    cs_o = cs_h = np.concatenate([obs, np.zeros(nu)])
    for _ in range(runs):
        state = np.random.get_state()
        uo, cs_o = orc.run(cs_o, obs)
        np.random.set_state(state)
        uh, cs_h = ctl.run(cs_h, obs, return_details=True)
        assert rel_err(ctl.last_costs, orc.last_costs) < tol
        assert rel_err(ctl.last_eps, orc.last_eps) < max(tol, 1e-12)
        assert rel_err(ctl.act_sequence, orc.act_sequence) < tol * 10
        assert rel_err(uh, uo) < tol * 10
        # keep the two in lock-step: only per-solve error is under test here
        ctl.act_sequence = orc.act_sequence
        obs = omodel.pred(obs, uo)


def test_mppi_multi_ctrl_small_vs_oracle():
    _oracle_vs_hip(17, 6, [256, 256], "relu", 300, 12, 1.0, 1.0, (-1.0, 1.0), "f64", 1e-9)


def test_mppi_multi_ctrl_tanh_per_particle_terminal():
    _oracle_vs_hip(5, 3, [100, 40], "tanh", 77, 9, 0.6, 0.4, (-0.7, 1.3), "f64", 1e-9, seed=3,
                   per_particle=True)


@pytest.mark.parametrize("nx,nu,hidden", [(40, 3, [256, 256]), (64, 8, [256, 128, 64]), (48, 6, [192, 192])])
def test_mppi_wide_states_with_wide_networks_vs_oracle(nx, nu, hidden):
    """More than 32 model states with hidden layers of up to 256 units (the reference's MLP configuration
    space is 16-256 units on any system, mlp.py:113-122): the WIDE tile (three or four output column tiles)."""
    _oracle_vs_hip(nx, nu, hidden, "tanh", 200, 8, 0.8, 0.9, (-1.0, 1.0), "f64", 1e-9, seed=nx)


def test_mppi_halfcheetah_full_size_vs_oracle():
    # BASELINE config 3: 4096 samples x 30 horizon, 17-dim state, 6 controls, MLP 2x256
    _oracle_vs_hip(17, 6, [256, 256], "relu", 4096, 30, 1.0, 1.0, (-1.0, 1.0), "f64", 1e-9, runs=1)


def test_mppi_pendulum_full_size_vs_oracle():
    # BASELINE config 2: 1024 x 30, 2-dim state, 1 control, MLP 2x64
    _oracle_vs_hip(2, 1, [64, 64], "relu", 1024, 30, 1.0, 1.0, (-2.0, 2.0), "f64", 1e-9, runs=2)


def test_mppi_f32_fast_mode_within_tolerance():
    # north_star tolerance: 1e-4 relative on state and cost
    _oracle_vs_hip(17, 6, [256, 256], "relu", 4096, 30, 1.0, 1.0, (-1.0, 1.0), "f32", 1e-4, runs=1)


def test_mppi_heterogeneous_batch_matches_single_solves():
    """The batch plan (different N, H, sigma, lmda, cost per problem) must give exactly what
    one-problem plans give: problems are independent."""
    from autompc_amd import _lib
    nx, nu = 17, 6
    p = omlp.random_params(nx, nu, [256, 256], "relu", seed=9)
    rng = np.random.default_rng(1)
    B = 5
    Ns, Hs = [100, 257, 64, 1000, 31], [5, 30, 17, 12, 8]
    sig, lam = rng.uniform(0.2, 2.0, size=B), rng.uniform(0.1, 2.0, size=B)
    Q = np.stack([np.diag(rng.uniform(0.1, 3.0, size=nx)) for _ in range(B)])
    R = np.stack([np.diag(rng.uniform(0.01, 0.2, size=nu)) for _ in range(B)])
    F = np.stack([np.diag(rng.uniform(0.1, 3.0, size=nx)) for _ in range(B)])
    goal = rng.normal(scale=0.1, size=(B, nx))
    h = _lib.Handle(0, "f64")
    h.set_mlp(nx, nu, p["weights"], p["biases"], "relu", p["xu_means"], p["xu_std"], p["dy_means"],
              p["dy_std"])
    h.set_quad_costs(Q, R, F, goal)
    h.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
    x0 = rng.uniform(-0.1, 0.1, size=(B, nx))
    acts = [rng.normal(size=(Hs[b], nu)) for b in range(B)]
    eps = [rng.normal(scale=np.sqrt(sig[b]), size=(Ns[b], Hs[b], nu)) for b in range(B)]
    plan = _lib.MppiPlan(h, Ns, Hs, sig, lam, cost_index=np.arange(B))
    plan.upload(x0, np.concatenate([a.ravel() for a in acts]), np.concatenate([e.ravel() for e in eps]))
    plan.solve()
    a_all, u_all, c_all, _ = plan.download(costs=True)
    ao = co = 0
    system = make_system(nx, nu)
    for b in range(B):
        orc = MPPIOracle(MLPOracle(system, p), QuadCostOracle(Q[b], R[b], F[b], goal[b]),
                         np.tile([-1.0, 1.0], (nu, 1)), horizon=Hs[b], num_path=Ns[b],
                         sigma=sig[b], lmda=lam[b])
        orc.act_sequence = acts[b].copy()
        u, _ = orc.run(np.concatenate([x0[b], np.zeros(nu)]), x0[b], eps_nhu=eps[b])
        assert rel_err(c_all[co:co + Ns[b]], orc.last_costs) < 1e-9
        assert rel_err(a_all[ao:ao + Hs[b] * nu].reshape(Hs[b], nu), orc.act_sequence) < 1e-8
        assert rel_err(u_all[b], u) < 1e-8
        ao += Hs[b] * nu
        co += Ns[b]


def test_device_noise_is_standard_normal():
    from autompc_amd import _lib
    nx, nu = 2, 1
    p = omlp.random_params(nx, nu, [64, 64], "relu", seed=1)
    h = _lib.Handle(0, "f64")
    h.set_mlp(nx, nu, p["weights"], p["biases"], "relu", p["xu_means"], p["xu_std"], p["dy_means"],
              p["dy_std"])
    h.set_quad_costs(np.eye(nx), 0.01 * np.eye(nu), np.eye(nx), np.zeros(nx))
    h.set_ctrl_bounds([-1.0], [1.0])
    # sigma = 0.01 -> std 0.1: with a zero warm start |eps| never reaches the clip at 1 (10 sigma),
    # so the post-clip noise the solve returns IS the generated noise.
    plan = _lib.MppiPlan(h, [20000], [30], [0.01], [1.0])
    plan.upload(np.zeros((1, nx)), np.zeros(30))
    plan.generate_eps(1234, 0)
    plan.solve()
    _, _, _, e = plan.download(eps_out=True)
    assert abs(e.mean()) < 1e-3 and abs(e.std() - 0.1) < 1e-3
    k = np.mean(((e - e.mean()) / e.std()) ** 4)
    assert abs(k - 3.0) < 0.05
    plan.generate_eps(1234, 1)
    plan.solve()
    _, _, _, e2 = plan.download(eps_out=True)
    assert abs(np.corrcoef(e, e2)[0, 1]) < 0.01
    plan.generate_eps(1234, 0)
    plan.solve()
    _, _, _, e3 = plan.download(eps_out=True)
    # counter-based: same (seed, stream) -> same noise (up to the (eps + a) - a round trip)
    np.testing.assert_allclose(e, e3, rtol=0, atol=1e-12)


def test_three_identical_controls_reduce_to_the_nu1_golden_on_device():
    """the nu > 1 path with identical per-dimension noise / weights / bounds / cost
    reproduces the reference's nu = 1 golden (construction: tests/test_nu_reduction.py)."""
    from autompc_amd import _lib
    from test_nu_reduction import reduction_problem
    g, nx, N, H, p3, act0, eps = reduction_problem()
    system, model, task = _hip_stack(p3, nx, 3, g["Q"], g["R"][0, 0] / 3.0 * np.eye(3), g["F"], g["goal"],
                                     g["bounds"])
    h = _lib.Handle(0, "f64")
    model.stage_into(h)
    h.set_quad_costs(g["Q"], g["R"][0, 0] / 3.0 * np.eye(3), g["F"], g["goal"])
    h.set_ctrl_bounds(np.full(3, g["bounds"][0]), np.full(3, g["bounds"][1]))
    plan = _lib.MppiPlan(h, [N], [H], [3.0 * float(g["sigma"])], [float(g["lmda"])])
    plan.upload(act_seq=act0)
    ref = MLPOracle(system, p3)
    obs = g["x0_0"].copy()
    for r in range(3):
        plan.upload(x0=obs, eps=eps[r])
        plan.solve()
        a, u, costs, _ = plan.download(costs=True)
        assert rel_err(costs, g["costs_%d" % r]) < 1e-9
        a = a.reshape(H, 3)
        for j in range(3):
            assert rel_err(a[:, j:j + 1], g["act_%d" % r]) < 1e-8
            assert rel_err(u[0, j:j + 1], g["u_%d" % r]) < 1e-8
        obs = ref.pred(obs, u[0])
    plan.close()
    h.close()


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_fused_update_with_an_all_inf_tile_matches_the_plain_update(precision, monkeypatch):
    """a rollout tile whose every sample has cost +inf (diverged, overflowing cost) must
    get weight 0 -- as in the reference's softmin (mppi.py:113-116) and in the non-fused update
    kernel -- instead of poisoning the sequence with exp(-(inf - inf)/lmda) = NaN."""
    from autompc_amd import _lib
    nx, nu, N, H = 17, 6, 96, 6
    p = omlp.random_params(nx, nu, [256, 256], "relu", seed=3)
    big = 1e300 if precision == "f64" else 1e34
    Q, R, F = big * np.eye(nx), 0.01 * np.eye(nu), np.eye(nx)
    system, model, task = _hip_stack(p, nx, nu, Q, R, F, np.zeros(nx), (-1e6, 1e6), precision)
    rng = np.random.default_rng(0)
    # noise and warm start are in units of umax = 1e6 and clipped to [-1, 1]: ordinary samples apply
    # controls of order 1, the samples of the third 16-row tile saturate at +-1e6, their states
    # reach ~1e5 and the stage cost big * x^2 overflows to +inf
    eps = 1e-6 * rng.normal(size=(N, H, nu))
    eps[32:48] = np.sign(rng.normal(size=(16, H, nu)))
    act0 = 1e-6 * rng.normal(size=(H, nu))
    x0 = rng.uniform(-0.1, 0.1, size=nx)
    out = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("AMPC_FUSED_UPDATE", fused)
        monkeypatch.setenv("AMPC_MT", "1")
        h = _lib.Handle(0, precision)
        model.stage_into(h)
        h.set_quad_costs(Q, R, F, np.zeros(nx))
        h.set_ctrl_bounds(np.full(nu, -1e6), np.full(nu, 1e6))
        plan = _lib.MppiPlan(h, [N], [H], [1.0], [1.0])
        plan.upload(x0=x0, act_seq=act0, eps=eps)
        plan.solve()
        a, u, costs, _ = plan.download(costs=True)
        out[fused] = (a, u, costs)
        plan.close()
        h.close()
    costs = out["1"][2]
    assert np.all(np.isinf(costs[32:48])) and np.all(np.isfinite(np.delete(costs, np.s_[32:48])))
    for fused in ("1", "0"):
        assert np.all(np.isfinite(out[fused][0])) and np.all(np.isfinite(out[fused][1]))
    tol = 1e-12 if precision == "f64" else 1e-5
    assert rel_err(out["1"][0], out["0"][0]) < tol and rel_err(out["1"][1], out["0"][1]) < tol


def test_act_sequence_in_place_edits_reach_the_device():
    """The reference's act_sequence is a plain attribute that callers edit in place
    (mppi.py:97-99).  After a solve the live copy is on the device: an in-place edit of the array
    the getter hands out must be uploaded before the next solve, exactly like an assignment."""
    from autompc_amd import MPPI
    nx = 2
    p = omlp.random_params(nx, 1, [64, 64], "relu", seed=2)
    system, model, task = _hip_stack(p, nx, 1, np.eye(nx), 0.01 * np.eye(1), np.eye(nx), np.zeros(nx), (-1, 1))
    results = []
    for mode in ("in_place", "slice", "setter"):
        np.random.seed(5)
        ctl = MPPI(system, task, model, horizon=5, num_path=64)
        ctl.run(np.zeros(3), np.zeros(2))
        if mode == "in_place":
            seq = ctl.act_sequence
            seq[:] = 0.0
            seq[1] += 0.25
        elif mode == "slice":
            ctl.act_sequence[:] = 0.0
            ctl.act_sequence[1, 0] = 0.25
        else:
            new = np.zeros((5, 1))
            new[1] = 0.25
            ctl.act_sequence = new
        expect = np.zeros((5, 1))
        expect[1] = 0.25
        np.testing.assert_array_equal(np.asarray(ctl.act_sequence), expect)
        u, _ = ctl.run(np.zeros(3), np.full(2, 0.1))
        results.append((u, np.asarray(ctl.act_sequence).copy()))
    for u, a in results[1:]:
        np.testing.assert_array_equal(u, results[0][0])
        np.testing.assert_array_equal(a, results[0][1])


@pytest.mark.parametrize("shape", [(17, 6, [256, 256], "relu"), (17, 6, [256, 256], "tanh"),
                                   (2, 1, [64, 64], "relu"), (4, 1, [64, 64], "selu"),
                                   (17, 6, [128, 128], "relu")])
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_shape_specialised_kernel_equals_the_general_kernel(shape, precision, monkeypatch):
    """Registered shapes (csrc/shapes.hpp) run mppi_rollout_kernel<..., StaticShape>; AMPC_STATIC=0
    forces the run-time-shape instantiation of the same kernel.  Same arithmetic in the same order:
    costs, clipped noise and the updated sequence must be bit-identical."""
    from autompc_amd import _lib
    nx, nu, hidden, act = shape
    N, H = 300, 12
    p = omlp.random_params(nx, nu, hidden, act, seed=11)
    rng = np.random.default_rng(3)
    Q, R, F = np.diag(rng.uniform(0.5, 2, nx)), np.diag(rng.uniform(0.01, 0.1, nu)), np.diag(rng.uniform(0.5, 2, nx))
    goal = rng.normal(scale=0.1, size=nx)
    system, model, task = _hip_stack(p, nx, nu, Q, R, F, goal, (-0.7, 0.9), precision)
    eps, act0 = rng.normal(size=(N, H, nu)), rng.normal(size=(H, nu))
    x0 = rng.uniform(-0.1, 0.1, size=nx)
    out = {}
    for static in ("1", "0"):
        monkeypatch.setenv("AMPC_STATIC", static)
        h = _lib.Handle(0, precision)
        model.stage_into(h)
        h.set_quad_costs(Q, R, F, goal)
        h.set_ctrl_bounds(np.full(nu, -0.7), np.full(nu, 0.9))
        plan = _lib.MppiPlan(h, [N, N // 2], [H, H - 3], [1.0, 0.6], [1.0, 0.4])
        plan.upload(x0=np.tile(x0, (2, 1)), act_seq=np.concatenate([act0.ravel(), act0[:H - 3].ravel()]),
                    eps=np.concatenate([eps.ravel(), eps[:N // 2, :H - 3].ravel()]))
        plan.solve()
        plan.solve()
        out[static] = plan.download(costs=True, eps_out=True)
        plan.close()
        h.close()
    for a, b in zip(out["1"], out["0"]):
        np.testing.assert_array_equal(a, b)
    # and both agree with the oracle on the first problem's costs of the second solve
    assert np.all(np.isfinite(out["1"][2]))


def test_a_view_held_across_run_never_resurrects_the_old_sequence():
    """the reference's act_sequence is ONE array mutated in place, so a reference kept
    across run() calls stays current.  Here the live copy moves to the device: a write through a view
    taken before run() must land on the CURRENT sequence (the device's warm start), not re-upload the
    pre-run one."""
    from autompc_amd import MPPI
    nx = 2
    p = omlp.random_params(nx, 1, [64, 64], "relu", seed=2)
    system, model, task = _hip_stack(p, nx, 1, np.eye(nx), 0.01 * np.eye(1), np.eye(nx), np.zeros(nx), (-1, 1))
    np.random.seed(5)
    ctl = MPPI(system, task, model, horizon=5, num_path=64)
    held = ctl.act_sequence                       # before any solve
    row = held[2]                                 # a sub-view of it
    before = np.asarray(held).copy()
    ctl.run(np.zeros(3), np.zeros(2))
    ctl.run(np.zeros(3), np.full(2, 0.05))
    live = np.asarray(ctl.act_sequence).copy()    # what the device holds now
    assert not np.array_equal(live, before)
    held[0, 0] = 0.5                              # write through the stale view
    expect = live.copy()
    expect[0, 0] = 0.5
    np.testing.assert_array_equal(np.asarray(ctl.act_sequence), expect)
    np.testing.assert_array_equal(np.asarray(held), expect)          # its memory was brought up to date
    row[0] = -0.25                                # and through the stale sub-view: no second refresh
    expect[2, 0] = -0.25
    np.testing.assert_array_equal(np.asarray(ctl.act_sequence), expect)
    # the next solve starts from exactly that sequence
    np.random.seed(9)
    u1, _ = ctl.run(np.zeros(3), np.full(2, 0.1))
    np.random.seed(5)
    ctl2 = MPPI(system, task, model, horizon=5, num_path=64)
    ctl2.act_sequence = expect
    np.random.seed(9)
    u2, _ = ctl2.run(np.zeros(3), np.full(2, 0.1))
    np.testing.assert_array_equal(u1, u2)


def test_noise_formed_one_solve_ahead_equals_the_generator_launch(monkeypatch):
    """Streams drawn in sequence (s, s + 1, ...): the combine launch of a solve forms the next stream index's
    Philox noise as well (ampc_mppi_plan::eps_next) and the predicted ampc_mppi_generate_eps is a pointer
    swap.  Same values as the generator launch, whatever the call pattern: sequences, jumps, repeated
    indices, changed noise ids, an uploaded buffer in between -- against AMPC_NOISE_AHEAD=0."""
    from autompc_amd import _lib
    nx, nu, H = 17, 6, 12
    p = omlp.random_params(nx, nu, [256, 256], "relu", seed=3)
    rng = np.random.default_rng(0)
    x0 = rng.uniform(-0.1, 0.1, size=(2, nx))
    upl = rng.normal(scale=0.3, size=(2 * 512 * H * nu))

    def run(ahead):
        monkeypatch.setenv("AMPC_NOISE_AHEAD", ahead)
        h = _lib.Handle(0, "f64")
        h.set_mlp(nx, nu, p["weights"], p["biases"], "relu", p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"])
        h.set_quad_costs(np.eye(nx), 0.01 * np.eye(nu), np.eye(nx), np.zeros(nx))
        h.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
        plan = _lib.MppiPlan(h, [512, 512], [H, H], [0.09, 0.04], [1.0, 0.5])     # sixteen-row tiles (two problems)
        plan.upload(x0, np.zeros(2 * H * nu))
        out = []

        def solve():
            plan.solve()
            a, u, _, e = plan.download(eps_out=True)
            out.append((a.copy(), u.copy(), e.copy()))
        for s in (5, 6, 7, 8, 8, 9, 20, 21, 22):              # sequence, a repeated index, a jump, a sequence again
            plan.generate_eps(77, s)
            solve()
        plan.set_noise_ids(np.array([11, 3], dtype=np.uint32))   # the ids key the stream: speculation is dropped
        for s in (23, 24):
            plan.generate_eps(77, s)
            solve()
        plan.upload(None, None, upl)                          # caller's own noise in between
        solve()
        for s in (25, 26, 27):
            plan.generate_eps(78 if s == 27 else 77, s)       # ... and a changed seed
            solve()
        plan.close(); h.close()
        return out
    a, b = run("1"), run("0")
    assert len(a) == len(b) == 15
    for i, (x, y) in enumerate(zip(a, b)):
        for k in range(3):
            assert np.array_equal(x[k], y[k]), (i, k)


def test_kernel_events_can_be_taken_on_every_nth_solve():
    """ampc_mppi_plan_set_timing(plan, n): the roofline leg's HIP events bracket every n-th solve only (three event
    records per solve cost ~11 us on the stream); the averages come from those solves, results are untouched."""
    from autompc_amd import _lib
    nx, nu, N, H = 17, 6, 512, 8
    p = omlp.random_params(nx, nu, [256, 256], "relu", seed=4)
    h = _lib.Handle(0, "f64")
    h.set_mlp(nx, nu, p["weights"], p["biases"], "relu", p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"])
    h.set_quad_costs(np.eye(nx), 0.1 * np.eye(nu), np.eye(nx), np.zeros(nx))
    h.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
    outs = []
    for every in (1, 4):
        plan = _lib.MppiPlan(h, [N], [H], [1.0], [1.0])
        plan.upload(np.zeros((1, nx)), np.zeros(H * nu))
        plan.set_timing(True, every=every)
        for i in range(8):
            plan.generate_eps(0, i)
            plan.solve()
        t = plan.timing()
        assert t["count"] == 8 // every and t["rollout_ms"] > 0 and t["update_ms"] > 0
        plan.set_timing(False)
        outs.append(plan.download(costs=True))
        plan.close()
    for a, b in zip(*outs):
        if a is not None:
            np.testing.assert_array_equal(a, b)
    h.close()


@pytest.mark.parametrize("shape", [(2, 1, [64, 64], [256, 100, 333], [12, 30, 7]), (17, 6, [256, 256], [512, 160], [10, 25])])
def test_one_call_step_with_several_problems_equals_upload_solve_download(shape, monkeypatch):
    """ampc_mppi_run on a plan of several problems (own N, H): x0 is read from mapped host memory, every problem's
    control is written there followed by ITS completion word (MppiArgs::done_flag), the host polls all of them.
    Controls and warm starts equal the step-by-step path (upload, generate_eps, solve, download) bit for bit, with
    and without mapped I/O, over consecutive steps; ampc_mppi_download after a one-call step returns its controls."""
    from autompc_amd import _lib
    nx, nu, hidden, Ns, Hs = shape
    p = omlp.random_params(nx, nu, hidden, "relu", seed=8)
    h = _lib.Handle(0, "f64")
    h.set_mlp(nx, nu, p["weights"], p["biases"], "relu", p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"])
    h.set_quad_costs(np.eye(nx), 0.01 * np.eye(nu), np.eye(nx), np.zeros(nx))
    h.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
    B = len(Ns)
    rng = np.random.default_rng(4)
    x0s = rng.uniform(-0.2, 0.2, size=(3, B, nx))
    act0 = rng.normal(scale=0.3, size=sum(Hs) * nu)
    outs = {}
    for mode in ("steps", "1", "0"):
        plan = _lib.MppiPlan(h, Ns, Hs, [0.6] * B, [0.8] * B)
        res = []
        for k in range(3):
            if mode == "steps":
                plan.upload(x0=x0s[k], act_seq=act0 if k == 0 else None)
                plan.generate_eps(5, k)
                plan.solve()
                a, u, _, _ = plan.download()
            else:
                monkeypatch.setenv("AMPC_RUN_MAPPED", mode)
                u = plan.run(x0s[k], act0 if k == 0 else None, philox=(5, k))
                a, u2, _, _ = plan.download()
                np.testing.assert_array_equal(u2, u)
            res.append((u.copy(), a.copy()))
        outs[mode] = res
        plan.close()
    for mode in ("1", "0"):
        for (u, a), (ur, ar) in zip(outs[mode], outs["steps"]):
            np.testing.assert_array_equal(u, ur)
            np.testing.assert_array_equal(a, ar)
    assert not np.array_equal(outs["1"][0][0][0], outs["1"][0][0][1])
    h.close()
