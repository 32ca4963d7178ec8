"""HIP MLP step / Jacobian vs the reference's golden vectors and the oracle (needs MI355X).
Every call goes through the C ABI (ampc_mlp_pred_batch / ampc_mlp_pred_diff_batch)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden
from helpers import check_weights, golden_params, make_system, rel_err
from oracle import mlp as omlp

pytestmark = pytest.mark.gpu

F64_TOL = 1e-10      # f64 MFMA vs f64 BLAS: only summation order differs
F32_TOL = 1e-4       # north_star tolerance for the f32 fast mode (measured ~1e-6)


def _names():
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "mlp_*.npz")))


def _hip_model(p, nx, nu, precision):
    from autompc_amd import MLP
    m = MLP(make_system(nx, nu), n_hidden_layers=len(p["weights"]) - 1,
            nonlintype=p["activation"], precision=precision,
            **{"hidden_size_%d" % (i + 1): w.shape[0] for i, w in enumerate(p["weights"][:-1])})
    m.weights = [w.copy() for w in p["weights"]]
    m.biases = [b.copy() for b in p["biases"]]
    m.xu_means, m.xu_std = p["xu_means"], p["xu_std"]
    m.dy_means, m.dy_std = p["dy_means"], p["dy_std"]
    return m


@pytest.mark.parametrize("precision,tol", [("f64", F64_TOL), ("f32", F32_TOL)])
@pytest.mark.parametrize("name", _names())
def test_pred_batch_and_jacobian_match_reference(name, precision, tol):
    g = golden(name)
    nx, nu = int(g["nx"]), int(g["nu"])
    p = golden_params(nx, nu, g["hidden"], g["activation"], g["seed"])
    check_weights(p, g)
    m = _hip_model(p, nx, nu, precision)
    out = m.pred_batch(g["states"], g["ctrls"])
    assert rel_err(out, g["pred_batch"]) < tol
    o2, jx, ju = m.pred_diff_batch(g["states"], g["ctrls"])
    assert rel_err(o2, g["diff_pred"]) < tol
    assert rel_err(jx, g["diff_jx"]) < tol * 10
    assert rel_err(ju, g["diff_ju"]) < tol * 10
    assert rel_err(m.pred(g["states"][0], g["ctrls"][0]), g["pred0"]) < tol
    o0, a0, b0 = m.pred_diff(g["states"][0], g["ctrls"][0])
    assert rel_err(a0, g["diff0_jx"]) < tol * 10 and rel_err(b0, g["diff0_ju"]) < tol * 10


@pytest.mark.parametrize("n", [1, 15, 16, 17, 100, 1000, 4096])
def test_ragged_batch_sizes_match_oracle(n):
    nx, nu = 17, 6
    p = omlp.random_params(nx, nu, [256, 256], "relu", seed=3)
    m = _hip_model(p, nx, nu, "f64")
    rng = np.random.default_rng(n)
    s, c = rng.normal(size=(n, nx)), rng.normal(size=(n, nu))
    assert rel_err(m.pred_batch(s, c), omlp.pred_batch(p, s, c)) < F64_TOL
    if n <= 100:
        o, jx, ju = m.pred_diff_batch(s, c)
        eo, ejx, eju = omlp.pred_diff_batch(p, s, c)
        assert rel_err(jx, ejx) < 1e-9 and rel_err(ju, eju) < 1e-9 and rel_err(o, eo) < F64_TOL


def test_set_parameters_restages_weights():
    nx, nu = 2, 1
    p1 = omlp.random_params(nx, nu, [64, 64], "tanh", seed=1)
    p2 = omlp.random_params(nx, nu, [64, 64], "tanh", seed=2)
    m = _hip_model(p1, nx, nu, "f64")
    s, c = np.ones((4, nx)) * 0.3, np.ones((4, nu)) * -0.2
    a = m.pred_batch(s, c)
    params = m.get_parameters()
    for k, (w, b) in zip(m._state_dict_keys(), zip(p2["weights"], p2["biases"])):
        params["net_state"][k[0]], params["net_state"][k[1]] = w, b
    m.set_parameters(params)
    b = m.pred_batch(s, c)
    assert rel_err(b, omlp.pred_batch(p2, s, c)) < F64_TOL and rel_err(a, b) > 1e-3


def test_empty_batch_returns_empty_arrays():
    """Zero rows in, zero rows out (numpy semantics of the reference's pred_batch on an empty
    batch): no launch, correctly shaped outputs."""
    nx, nu = 5, 2
    m = _hip_model(omlp.random_params(nx, nu, [64, 64], "tanh", seed=2), nx, nu, "f64")
    out = m.pred_batch(np.zeros((0, nx)), np.zeros((0, nu)))
    assert out.shape == (0, nx)
    o, jx, ju = m.pred_diff_batch(np.zeros((0, nx)), np.zeros((0, nu)))
    assert o.shape == (0, nx) and jx.shape == (0, nx, nx) and ju.shape == (0, nx, nu)
    with pytest.raises(ValueError):
        m.pred_batch(np.zeros((3, nx + 1)), np.zeros((3, nu)))
