"""The four-row MPPI rollout kernel (csrc/mppi_rollout4.hpp; tile_rows = 4) against the sixteen-row
kernel and the CPU oracle: same contract (mppi.py:110-152), another tiling.  Needs MI355X."""
import numpy as np
import pytest

from helpers import make_system, rel_err
from oracle import mlp as omlp
from oracle.costs import QuadCostOracle
from oracle.mppi import MPPIOracle

pytestmark = pytest.mark.gpu


def _handle(nx, nu, hidden, act, dense, seed=3):
    from autompc_amd import MLP, _lib
    system = make_system(nx, nu)
    p = omlp.random_params(nx, nu, hidden, act, seed=seed)
    kw = {"hidden_size_%d" % (i + 1): h for i, h in enumerate(hidden)}
    m = MLP(system, n_hidden_layers=len(hidden), nonlintype=act, **kw)
    m.weights, m.biases = p["weights"], p["biases"]
    m.xu_means, m.xu_std, m.dy_means, m.dy_std = p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"]
    h = _lib.Handle(0, "f64")
    m.stage_into(h)
    rng = np.random.default_rng(seed)
    if dense:
        A = rng.normal(size=(nx, nx)); Q = A @ A.T / nx + np.eye(nx)
        Bm = rng.normal(size=(nu, nu)); R = 0.05 * (Bm @ Bm.T / nu + np.eye(nu))
        C = rng.normal(size=(nx, nx)); F = C @ C.T / nx + np.eye(nx)
    else:
        Q, R, F = np.diag(rng.uniform(0.5, 2, nx)), np.diag(rng.uniform(0.01, 0.1, nu)), np.diag(rng.uniform(0.5, 3, nx))
    goal = rng.normal(size=nx) * 0.1
    h.set_quad_costs(Q, R, F, goal)
    lo, hi = -rng.uniform(0.3, 1.0, nu), rng.uniform(0.5, 1.0, nu)
    h.set_ctrl_bounds(lo, hi)
    return system, p, h, (Q, R, F, goal, lo, hi)


def _solve(h, rows, N, H, sigma, lmda, x0, act, eps, term_mode):
    from autompc_amd import _lib
    plan = _lib.MppiPlan(h, N, H, sigma, lmda, term_mode=term_mode)
    plan.set_geometry(rows, 0)
    plan.upload(x0=x0, act_seq=act, eps=eps)
    plan.solve()
    out = plan.download(costs=True, eps_out=True)
    info, kind = plan.info(), plan.kernel_kind()
    plan.close()
    return out, info, kind


@pytest.mark.parametrize("nx,nu,hidden,act,dense,term", [
    (2, 1, [64, 64], "relu", False, 0),            # BASELINE config 2's shape (registered: specialised kernel)
    (4, 1, [64, 64], "tanh", False, 1),            # registered, run-time activation
    (3, 2, [32], "relu", True, 0),                 # one hidden layer, dense cost blocks
    (5, 3, [64, 48, 64], "sigmoid", False, 0),     # three hidden layers
    (7, 2, [16, 32, 64, 16], "selu", True, 1),     # four hidden layers
    (20, 6, [128, 128], "relu", False, 0),         # hpad 128, two output tiles (k-split output layer)
    (11, 4, [100], "tanh", False, 0),              # hpad 128, one output tile
    (30, 2, [64, 64], "relu", True, 0),            # 32 states: two output tiles, every wave the whole layer
])
def test_four_row_kernel_equals_sixteen_row_kernel_and_oracle(nx, nu, hidden, act, dense, term):
    system, p, h, (Q, R, F, goal, lo, hi) = _handle(nx, nu, hidden, act, dense)
    rng = np.random.default_rng(11)
    N, H = np.array([37, 130]), np.array([9, 6])          # two problems, ragged tiles (37 = 9 * 4 + 1)
    sigma, lmda = np.array([0.3, 0.6]), np.array([0.7, 1.3])
    x0 = rng.normal(size=(2, nx)) * 0.2
    act_seq = rng.uniform(-0.3, 0.3, size=int(np.sum(H * nu)))
    eps = np.concatenate([rng.normal(size=N[b] * H[b] * nu) * np.sqrt(sigma[b]) for b in range(2)])
    (a4, u4, c4, e4), info4, kind4 = _solve(h, 4, N, H, sigma, lmda, x0, act_seq, eps, term)
    (a16, u16, c16, e16), info16, _ = _solve(h, 16, N, H, sigma, lmda, x0, act_seq, eps, term)
    assert info4["samples_per_wg"] == 4 and info4["workgroups"] == 10 + 33
    assert info16["samples_per_wg"] == 16
    assert kind4 in (1, 2, 3)
    np.testing.assert_array_equal(e4, e16)               # clipped noise: the same elementwise arithmetic
    assert rel_err(c4, c16) < 1e-12 and rel_err(a4, a16) < 1e-11 and rel_err(u4, u16) < 1e-11
    # ... and the CPU restatement of the reference, problem by problem
    model = omlp.MLPOracle(system, p)
    off_a = off_e = off_c = 0
    for b in range(2):
        n, hh = int(N[b]), int(H[b])
        np.random.seed(0)
        orc = MPPIOracle(model, QuadCostOracle(Q, R, F, goal), np.stack([lo, hi], axis=1), horizon=hh, num_path=n,
                         sigma=float(sigma[b]), lmda=float(lmda[b]), per_particle_terminal=bool(term))
        orc.act_sequence = act_seq[off_a:off_a + hh * nu].reshape(hh, nu).copy()
        costs, e = orc.do_rollouts(x0[b], eps[off_e:off_e + n * hh * nu].reshape(n, hh, nu))
        orc.update(costs, e)
        assert rel_err(c4[off_c:off_c + n], costs) < 1e-10
        assert rel_err(a4[off_a:off_a + hh * nu].reshape(hh, nu), orc.act_sequence) < 1e-9
        off_a += hh * nu; off_e += n * hh * nu; off_c += n
    h.close()


@pytest.mark.parametrize("nx,nu,hidden,act", [(2, 1, [64, 64], "relu"), (4, 1, [64, 64], "tanh"), (17, 6, [128, 128], "relu")])
def test_specialised_and_runtime_shape_versions_give_the_same_bits(nx, nu, hidden, act, monkeypatch):
    """Registered shapes run the kernel with dimensions and activation folded at compile time
    (AMPC_STATIC=0: the run-time-shape version).  Same operations in the same order: same bits."""
    system, p, h, _ = _handle(nx, nu, hidden, act, False)
    rng = np.random.default_rng(5)
    N, H = np.array([203]), np.array([12])
    x0 = rng.normal(size=(1, nx)) * 0.2
    act_seq = rng.uniform(-0.3, 0.3, size=12 * nu)
    eps = rng.normal(size=203 * 12 * nu) * 0.5
    out = {}
    for static in ("1", "0"):
        monkeypatch.setenv("AMPC_STATIC", static)
        out[static], _, kind = _solve(h, 4, N, H, [0.25], [0.9], x0, act_seq, eps, 0)
        assert kind == (1 if static == "1" else 3)
    for a, b in zip(out["1"], out["0"]):
        np.testing.assert_array_equal(a, b)
    h.close()


def test_automatic_choice_and_refusals(monkeypatch):
    from autompc_amd import _lib
    monkeypatch.delenv("AMPC_QUAD", raising=False)        # (the automatic rule is what is under test)
    from autompc_amd._lib import AmpcError
    system, p, h, _ = _handle(2, 1, [64, 64], "relu", False)
    small = _lib.MppiPlan(h, [1024], [30], [1.0], [1.0])          # 64 sixteen-row tiles on 256 CUs
    assert small.info()["samples_per_wg"] == 4
    big = _lib.MppiPlan(h, [4096], [30], [1.0], [1.0])            # enough rows to fill the chip
    assert big.info()["samples_per_wg"] >= 16
    big.set_geometry(4, 0)                                         # ... unless the caller insists
    assert big.info()["samples_per_wg"] == 4
    small.close(); big.close(); h.close()
    # shapes the four-row kernel does not cover: forcing it is an error, not a silent fallback
    system, p, h, _ = _handle(17, 6, [256, 256], "relu", False)
    plan = _lib.MppiPlan(h, [64], [10], [1.0], [1.0])
    assert plan.info()["samples_per_wg"] >= 16
    with pytest.raises(AmpcError):
        plan.set_geometry(4, 0)
    plan.close(); h.close()
    h32 = _lib.Handle(0, "f32")
    from autompc_amd import MLP
    m = MLP(make_system(2, 1), n_hidden_layers=2, hidden_size=64, nonlintype="relu", precision="f32")
    q = omlp.random_params(2, 1, [64, 64], "relu", seed=1)
    m.weights, m.biases = q["weights"], q["biases"]
    m.xu_means, m.xu_std, m.dy_means, m.dy_std = q["xu_means"], q["xu_std"], q["dy_means"], q["dy_std"]
    m.stage_into(h32)
    h32.set_quad_costs(np.eye(2), 0.01 * np.eye(1), np.eye(2), np.zeros(2))
    h32.set_ctrl_bounds([-1.0], [1.0])
    plan = _lib.MppiPlan(h32, [256], [10], [1.0], [1.0])
    assert plan.info()["samples_per_wg"] >= 16                    # (f64 only)
    plan.close(); h32.close()


@pytest.mark.parametrize("nx,nu,hidden,N,H", [(2, 1, [64, 64], 1024, 30), (5, 3, [64, 48], 333, 11), (11, 4, [100], 50, 7)])
def test_noise_formed_inside_the_rollout_equals_the_generator_kernel(nx, nu, hidden, N, H, monkeypatch):
    """Device Philox noise on a four-row plan is formed in the rollout's prologue (no generator
    launch, no buffer round trip): every value must be the one philox_normal_batch_kernel writes --
    costs, clipped noise, updated sequence and control bit for bit, for odd element counts and ragged
    tiles, across consecutive streams, and when a solve is repeated on the same noise."""
    from autompc_amd import _lib
    system, p, h, _ = _handle(nx, nu, hidden, "tanh", False)
    rng = np.random.default_rng(N)
    x0, act = rng.uniform(-0.2, 0.2, size=nx), rng.normal(size=H * nu)
    outs = {}
    for inline in ("0", "1"):
        monkeypatch.setenv("AMPC_INLINE_NOISE", inline)
        plan = _lib.MppiPlan(h, N, H, 0.8, 0.6)
        plan.set_geometry(4, 0)
        plan.set_noise_ids(np.array([7], dtype=np.uint32))
        plan.upload(x0=x0, act_seq=act)
        res = []
        for stream in (0, 1, 2 ** 33 + 5):
            plan.generate_eps(11, stream)
            plan.solve()
            res.append(plan.download(costs=True, eps_out=True))
        plan.solve()                                   # again on the same noise: the stream did not move
        res.append(plan.download(costs=True, eps_out=True))
        u = plan.run(x0, act, philox=(11, 3))          # the drop-in classes' one-call form
        res.append((u,))
        assert plan.kernel_kind() in (1, 3)
        outs[inline] = res
        plan.close()
    for a, b in zip(outs["0"], outs["1"]):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    # and the repeated solve saw the same noise as the one before it (same clipped noise where unclipped)
    np.testing.assert_array_equal(outs["1"][3][3] != 0, outs["1"][2][3] != 0)
    h.close()
