"""Size-independent properties of the MPPI solve at the full BASELINE size (C3: 17-dim state,
6 controls, MLP 2x256, 4096 samples x 30 horizon), checked through the C ABI without any
reference run: what the algorithm (mppi.py:110-152) guarantees whatever the numbers are."""
import numpy as np
import pytest

from helpers import rel_err

pytestmark = pytest.mark.gpu

NX, NU, N, H = 17, 6, 4096, 30


def _plan(lmda=1.0, scale=1.0, per_particle=False):
    from autompc_amd import _lib
    from autompc_amd.synthetic import make_workload
    system, task, model, spec = make_workload("c3")
    h = _lib.Handle(0, "f64")
    model.stage_into(h)
    Q, R, F = task.get_cost().get_cost_matrices()
    h.set_quad_costs(scale * Q, scale * R, scale * F, task.get_cost().get_goal())
    b = task.get_ctrl_bounds()
    h.set_ctrl_bounds(b[:, 0], b[:, 1])
    plan = _lib.MppiPlan(h, [N], [H], [1.0], [lmda],
                         term_mode=_lib.TERM_PER_PARTICLE if per_particle else _lib.TERM_REFERENCE)
    return h, plan, task.get_init_obs()


def _solve(plan, x0, act, eps):
    plan.upload(x0, act.ravel(), eps.ravel())
    plan.solve()
    a, u, c, e = plan.download(costs=True, eps_out=True)
    return a.reshape(H, NU), u[0], c, e.reshape(H, N, NU)


def test_zero_noise_returns_the_shifted_sequence_exactly():
    """eps = 0: every particle follows the warm start, all costs are equal, the softmin weights are
    uniform and the update adds sum_n w_n * 0 -- the result is the shifted sequence, bit for bit
    (mppi.py:121-123: a[:-1] = a[1:], last entry kept)."""
    h, plan, x0 = _plan()
    act = np.random.default_rng(0).uniform(-0.5, 0.5, size=(H, NU))
    a, u, c, e = _solve(plan, x0, act, np.zeros((N, H, NU)))
    shifted = np.concatenate([act[1:], act[-1:]])
    np.testing.assert_array_equal(a, shifted)
    np.testing.assert_array_equal(u, shifted[0] * 1.0)          # umax = 1
    assert np.all(c == c[0]) and np.all(e == 0.0)
    plan.close(); h.close()


def test_particle_permutation_permutes_costs_and_keeps_the_update():
    h, plan, x0 = _plan(per_particle=True)
    rng = np.random.default_rng(1)
    act = rng.uniform(-0.3, 0.3, size=(H, NU))
    eps = rng.normal(size=(N, H, NU))
    perm = rng.permutation(N)
    a1, u1, c1, _ = _solve(plan, x0, act, eps)
    a2, u2, c2, _ = _solve(plan, x0, act, eps[perm])
    np.testing.assert_array_equal(c2, c1[perm])                 # a sample's cost does not depend on its slot
    assert rel_err(a2, a1) < 1e-12 and rel_err(u2, u1) < 1e-12  # only the summation order differs
    plan.close(); h.close()


def test_scaling_costs_and_temperature_together_leaves_the_update_unchanged():
    """softmin(c / lmda) is invariant under (c, lmda) -> (s c, s lmda); the action-cost term
    lmda/sigma * a.eps scales with it (mppi.py:113-118, 143)."""
    rng = np.random.default_rng(2)
    act = rng.uniform(-0.3, 0.3, size=(H, NU))
    eps = rng.normal(size=(N, H, NU))
    h1, p1, x0 = _plan(lmda=1.0, scale=1.0)
    a1, u1, c1, e1 = _solve(p1, x0, act, eps)
    h2, p2, _ = _plan(lmda=4.0, scale=4.0)
    a2, u2, c2, e2 = _solve(p2, x0, act, eps)
    assert rel_err(c2, 4.0 * c1) < 1e-13                         # power-of-two scale: exact up to fma order
    np.testing.assert_array_equal(e2, e1)
    assert rel_err(a2, a1) < 1e-12 and rel_err(u2, u1) < 1e-12
    for x in (p1, p2): x.close()
    for x in (h1, h2): x.close()


def test_clipped_noise_respects_bounds_and_solve_is_repeatable():
    h, plan, x0 = _plan()
    rng = np.random.default_rng(3)
    act = rng.uniform(-0.9, 0.9, size=(H, NU))
    eps = rng.normal(size=(N, H, NU)) * 2.0
    a1, u1, c1, e1 = _solve(plan, x0, act, eps)
    shifted = np.concatenate([act[1:], act[-1:]])
    applied = e1 + shifted[:, None, :]                           # eps <- clip(a + eps) - a  (mppi.py:134-139)
    assert applied.min() >= -1.0 - 1e-15 and applied.max() <= 1.0 + 1e-15
    # the stored noise is (a + eps) - a, rounded the way the reference rounds it (mppi.py:137-138)
    raw = np.transpose(eps, (1, 0, 2)) + shifted[:, None, :]
    np.testing.assert_array_equal(e1, np.clip(raw, -1.0, 1.0) - shifted[:, None, :])
    a2, u2, c2, e2 = _solve(plan, x0, act, eps)
    np.testing.assert_array_equal(a2, a1)
    np.testing.assert_array_equal(c2, c1)
    plan.close(); h.close()


def test_ilqr_on_linear_dynamics_recovers_finite_horizon_lqr():
    """On x' = A x + B u with a quadratic cost the iLQR step is an exact Newton step: the feedback
    gains must equal the finite-horizon discrete LQR gains (closed-form backward Riccati in numpy),
    the solve must converge in a handful of iterations and the trajectory must follow u = K x."""
    from autompc_amd import _lib
    rng = np.random.default_rng(5)
    nx, nu, Hh, dt = 12, 3, 50, 0.05
    S = rng.normal(size=(nx, nx))
    A = np.eye(nx) + 0.1 * (-0.3 * np.eye(nx) + 0.4 * (S - S.T))
    B = rng.normal(scale=0.3, size=(nx, nu))
    Q = np.diag(rng.uniform(0.5, 2.0, size=nx))
    R = np.diag(rng.uniform(0.05, 0.2, size=nu))
    F = np.diag(rng.uniform(1.0, 3.0, size=nx))
    h = _lib.Handle(0, "f64")
    h.set_linear(A, B)
    h.set_quad_costs(Q, R, F, np.zeros(nx))
    plan = _lib.IlqrPlan(h, 1, Hh, dt)
    x0 = rng.uniform(-1.0, 1.0, size=nx)
    out = plan.solve(x0[None, :], np.zeros((1, Hh, nu)), 50)
    assert out["status"][0] == 0 and out["converged"][0] == 1 and out["iters"][0] <= 4
    # closed form: V_H = 2F;  K_t = -(2 R dt + B'VB)^-1 B'VA;  V <- 2 Q dt + A'VA + A'VB K_t
    V = 2.0 * F
    Ks = np.zeros((Hh, nu, nx))
    for t in range(Hh - 1, -1, -1):
        G = 2.0 * R * dt + B.T @ V @ B
        Ks[t] = -np.linalg.solve(G, B.T @ V @ A)
        V = 2.0 * Q * dt + A.T @ V @ A + A.T @ V @ B @ Ks[t]
    assert rel_err(out["Ks"][0], Ks) < 1e-9
    x = x0.copy()
    for t in range(Hh):
        u = Ks[t] @ x
        assert np.max(np.abs(out["ctrls"][0, t] - u)) < 1e-8 * max(1.0, np.max(np.abs(u)))
        assert np.max(np.abs(out["states"][0, t] - x)) < 1e-8
        x = A @ x + B @ u
    plan.close(); h.close()


def test_ilqr_on_a_wide_linear_model_recovers_lqr_too():
    """41 states + 6 controls (ARX history 2 on a HalfCheetah-sized system): the model runs on the
    four-output-tile MFMA path, the Riccati sweep holds 47 x 47 matrices in LDS and the Quu solve
    uses 48 lanes; the gains must still be the closed-form finite-horizon LQR gains."""
    from autompc_amd import _lib
    rng = np.random.default_rng(9)
    nx, nu, no, Hh, dt = 41, 6, 17, 20, 0.05
    S = rng.normal(size=(nx, nx))
    A = 0.95 * np.eye(nx) + 0.05 * (S - S.T) / np.sqrt(nx)
    B = rng.normal(scale=0.2, size=(nx, nu))
    Q = np.diag(rng.uniform(0.5, 2.0, size=no))
    R = np.diag(rng.uniform(0.05, 0.2, size=nu))
    F = np.diag(rng.uniform(1.0, 3.0, size=no))
    h = _lib.Handle(0, "f64")
    h.set_linear(A, B)
    h.set_quad_costs(Q, R, F, np.zeros(no))
    plan = _lib.IlqrPlan(h, 2, Hh, dt)
    x0 = rng.uniform(-1.0, 1.0, size=(2, nx))
    out = plan.solve(x0, np.zeros((2, Hh, nu)), 50)
    assert np.all(out["status"] == 0) and np.all(out["converged"] == 1) and np.all(out["iters"] <= 4)
    Qx, Fx = np.zeros((nx, nx)), np.zeros((nx, nx))
    Qx[:no, :no], Fx[:no, :no] = Q, F               # the cost sees the first obs_dim state entries
    V = 2.0 * Fx
    Ks = np.zeros((Hh, nu, nx))
    for t in range(Hh - 1, -1, -1):
        G = 2.0 * R * dt + B.T @ V @ B
        Ks[t] = -np.linalg.solve(G, B.T @ V @ A)
        V = 2.0 * Qx * dt + A.T @ V @ A + A.T @ V @ B @ Ks[t]
    for b in range(2):
        assert rel_err(out["Ks"][b], Ks) < 1e-8
    plan.close(); h.close()
