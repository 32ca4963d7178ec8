"""numpy's legacy normal stream generated on the device (ampc_mppi_legacy_normal) against numpy
itself: the reference draws MPPI's noise with np.random.normal(scale=sqrt(sigma), size=(N, H, nu))
from the global legacy generator (autompc/control/mppi.py:16-24, :126).  Needs MI355X."""
import numpy as np
import pytest

from helpers import check_weights, golden_params, make_system, rel_err
from conftest import golden
from oracle import mlp as omlp

pytestmark = pytest.mark.gpu


def _plan(N, H, sigma, nx=2, nu=1, precision="f64", B=1):
    from autompc_amd import MLP, _lib
    system = make_system(nx, nu)
    p = omlp.random_params(nx, nu, [64, 64], "relu", seed=1)
    m = MLP(system, n_hidden_layers=2, hidden_size=64, nonlintype="relu", precision=precision)
    m.weights, m.biases = p["weights"], p["biases"]
    m.xu_means, m.xu_std, m.dy_means, m.dy_std = p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"]
    h = _lib.Handle(0, precision)
    m.stage_into(h)
    h.set_quad_costs(np.eye(nx), 0.01 * np.eye(nu), np.eye(nx), np.zeros(nx))
    h.set_ctrl_bounds(-np.ones(nu), np.ones(nu))     # (tests use sigma << 1: nothing reaches the clip at +-1)
    N, H, sigma = np.atleast_1d(N), np.atleast_1d(H), np.atleast_1d(sigma)
    plan = _lib.MppiPlan(h, N, H, sigma, np.ones(len(N)))
    return h, plan


def _exact():
    """The library has proven its restatement of the host's log() (ampc_legacy_log_mode != 0): the
    device-generated normals must then be numpy's bit for bit."""
    from autompc_amd import _lib
    return _lib.legacy_log_mode() != 0


def test_host_log_is_reproduced_on_this_box():
    """The GPU boxes run glibc 2.35 on FMA-capable CPUs: the exact path is the one under test."""
    from autompc_amd import _lib
    assert _lib.legacy_log_mode() in (1, 2)


def test_ten_million_normals_are_numpys_bit_for_bit():
    """14 consecutive config-3 draws (4096 x 30 x 6 = 737 280 values each, 10.3 M in all) against
    np.random.normal from the same generator state: array_equal, and the generator state handed
    back is numpy's after every call."""
    if not _exact():
        pytest.skip("the host's log() is not one of the two glibc builds the library reproduces")
    N, H, nu, sigma = 4096, 30, 6, 0.0049
    h, plan = _plan(N, H, sigma, nx=2, nu=nu)
    np.random.seed(1234)
    total = 0
    for call in range(14):
        st0 = np.random.get_state()
        ref = np.random.normal(scale=np.sqrt(sigma), size=(N, H, nu))
        st_ref = np.random.get_state()
        np.random.set_state(st0)
        st_dev, e = _device_draw(plan)
        np.testing.assert_array_equal(e.reshape(H, N, nu).transpose(1, 0, 2), ref)
        np.testing.assert_array_equal(st_dev[1], st_ref[1])
        assert st_dev[2:] == st_ref[2:]
        np.random.set_state(st_dev)
        total += ref.size
    assert total >= 10_000_000
    plan.close()
    h.close()


def _device_draw(plan):
    """legacy_normal + read the noise back through a solve with a zero warm start and noise far
    inside the bounds (eps_out == eps, laid out [H][N][nu] per problem)."""
    new_state = plan.legacy_normal(np.random.get_state())
    plan.upload(x0=np.zeros((plan.B, plan.handle.nx)), act_seq=np.zeros(plan.sum_hnu))
    plan.solve()
    _, _, _, e = plan.download(act_seq=False, u=False, eps_out=True)
    return new_state, e


@pytest.mark.parametrize("N,H,nu,seed,pre", [(1000, 20, 1, 0, 0), (333, 7, 3, 5, 0), (64, 5, 1, 11, 1),
                                             (4096, 30, 6, 3, 0), (50, 3, 1, 2, 3)])
def test_stream_and_state_match_numpy(N, H, nu, seed, pre):
    """Same draw, same generator state afterwards.  `pre` normals are drawn on the host first so
    that the call starts with a value in numpy's cache (odd counts) / mid-block positions."""
    sigma = 0.0049            # std 0.07: 14 sigma to the clip
    h, plan = _plan(N, H, sigma, nx=2, nu=nu)
    np.random.seed(seed)
    np.random.normal(size=pre)
    st0 = np.random.get_state()
    ref = np.random.normal(scale=np.sqrt(sigma), size=(N, H, nu))
    st_ref = np.random.get_state()
    np.random.set_state(st0)
    st_dev, e = _device_draw(plan)
    got = e.reshape(H, N, nu).transpose(1, 0, 2)
    # every accept / reject decision and the MT19937 state are exact ...
    assert st_dev[2] == st_ref[2] and st_dev[3] == st_ref[3]
    np.testing.assert_array_equal(st_dev[1], st_ref[1])
    if _exact():
        # ... and so are the normals: the device evaluates the host C library's log() itself
        if st_ref[3]:
            assert st_dev[4] == st_ref[4]
        np.testing.assert_array_equal(got, ref)
    else:
        # ... the normals agree to the last bit or two (device log vs an unknown host libm)
        if st_ref[3]:
            assert abs(st_dev[4] - st_ref[4]) <= 2 * np.spacing(abs(st_ref[4]))
        ulp = np.spacing(np.abs(ref))
        assert np.max(np.abs(got - ref) / ulp) <= 4.0
        assert np.mean(got == ref) > 0.95
    # the host generator continues exactly where numpy's own draw would have left it
    np.random.set_state(st_dev)
    a = np.random.random_sample(5)
    np.random.set_state(st_ref)
    np.testing.assert_array_equal(a, np.random.random_sample(5))
    plan.close()
    h.close()


def test_two_problems_draw_in_turn_with_their_own_scale():
    h, plan = _plan([40, 25], [6, 4], [0.005, 0.02], nx=2, nu=1)
    np.random.seed(9)
    st0 = np.random.get_state()
    r0 = np.random.normal(scale=np.sqrt(0.005), size=(40, 6, 1))
    r1 = np.random.normal(scale=np.sqrt(0.02), size=(25, 4, 1))
    np.random.set_state(st0)
    _, e = _device_draw(plan)
    g0 = e[:240].reshape(6, 40, 1).transpose(1, 0, 2)
    g1 = e[240:].reshape(4, 25, 1).transpose(1, 0, 2)
    if _exact():
        np.testing.assert_array_equal(g0, r0)
        np.testing.assert_array_equal(g1, r1)
    assert rel_err(g0, r0) < 1e-15 and rel_err(g1, r1) < 1e-15
    plan.close()
    h.close()


@pytest.mark.parametrize("noise", ["numpy_device", None])
@pytest.mark.parametrize("name", ["mppi_c2_pendulum", "mppi_clip_asym", "mppi_hc_nu1", "mppi_lowlmda_goal"])
def test_mppi_golden_with_device_drawn_numpy_noise(name, noise):
    """The reference's golden MPPI runs reproduced with the draw made on the device: same seeds;
    the warm start (drawn on the host at construction) is bit-identical.  noise=None is the
    DEFAULT mode ("numpy"), which takes the device path whenever it is provably exact."""
    from autompc_amd import MLP, MPPI, QuadCost, Task
    from oracle.mlp import MLPOracle
    g = golden(name)
    nx = int(g["nx"])
    p = golden_params(nx, 1, g["hidden"], g["activation"], g["mlp_seed"], bool(g["plain_norm"]))
    check_weights(p, g)
    system = make_system(nx, 1)
    m = MLP(system, n_hidden_layers=len(p["weights"]) - 1, nonlintype=p["activation"],
            **{"hidden_size_%d" % (i + 1): w.shape[0] for i, w in enumerate(p["weights"][:-1])})
    m.weights, m.biases = p["weights"], p["biases"]
    m.xu_means, m.xu_std, m.dy_means, m.dy_std = p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"]
    task = Task(system)
    task.set_cost(QuadCost(system, g["Q"], g["R"], g["F"], goal=g["goal"]))
    task.set_ctrl_bounds([g["bounds"][0]], [g["bounds"][1]])
    np.random.seed(int(g["np_seed"]))
    kw = {} if noise is None else {"noise": noise}
    ctl = MPPI(system, task, m, horizon=int(g["H"]), num_path=int(g["N"]), sigma=float(g["sigma"]),
               lmda=float(g["lmda"]), **kw)
    if noise is None:
        assert ctl.noise == "numpy"
    np.testing.assert_array_equal(ctl.act_sequence, g["act0"])
    obs = np.random.default_rng(int(g["np_seed"]) + 99).uniform(-0.1, 0.1, size=nx)
    constate = np.concatenate([obs, np.zeros(1)])
    ref_model = MLPOracle(system, p)
    for r in range(int(g["n_runs"])):
        if r == 3:
            ctl.reset()
            np.testing.assert_array_equal(ctl.act_sequence, g["act_reset"])   # host stream stayed in step
        u, constate = ctl.run(constate, obs, return_details=True)
        assert rel_err(ctl.last_costs, g["costs_%d" % r]) < 1e-9
        assert rel_err(ctl.last_eps[:, ::16, :], g["eps_sub_%d" % r]) < 1e-12
        assert rel_err(ctl.act_sequence, g["act_%d" % r]) < 1e-8
        assert rel_err(u, g["u_%d" % r]) < 1e-8
        obs = ref_model.pred(obs, g["u_%d" % r])


def test_consecutive_calls_and_interleaved_host_draws():
    """The next call's raw stream is generated speculatively from the state a call leaves behind:
    consecutive calls take that path, a host draw in between invalidates it -- both must continue
    numpy's stream exactly."""
    N, H, nu, sigma = 500, 10, 2, 0.0049
    h, plan = _plan(N, H, sigma, nx=2, nu=nu)
    np.random.seed(21)
    for call in range(6):
        if call in (3, 5):
            np.random.normal(size=3 + call)          # someone else draws: speculation misses
        st0 = np.random.get_state()
        ref = np.random.normal(scale=np.sqrt(sigma), size=(N, H, nu))
        st_ref = np.random.get_state()
        np.random.set_state(st0)
        st_dev, e = _device_draw(plan)
        got = e.reshape(H, N, nu).transpose(1, 0, 2)
        assert st_dev[2] == st_ref[2] and st_dev[3] == st_ref[3]
        np.testing.assert_array_equal(st_dev[1], st_ref[1])
        assert np.max(np.abs(got - ref) / np.spacing(np.abs(ref))) <= 4.0
        if _exact():
            np.testing.assert_array_equal(got, ref)
        np.random.set_state(st_dev)
    plan.close()
    h.close()


_FORCED_BUILD_SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
from oracle import glibc_log
from autompc_amd import _lib
import test_gpu_legacy_noise as T
build = {build}
assert _lib.legacy_log_mode() == build
table = glibc_log.locate()
N, H, nu, sigma = 2048, 20, 3, 0.0049
h, plan = T._plan(N, H, sigma, nx=2, nu=nu)
np.random.seed(77)
st0 = np.random.get_state()
rs = np.random.RandomState()
rs.set_state(st0)
# numpy's legacy_gauss, with log() replaced by the restated build: pairs of uniforms -> polar
# method, accepted pairs in order; value 2k is f * x2, value 2k + 1 is f * x1
need = N * H * nu
u = rs.random_sample(4 * need)
x1, x2 = 2.0 * u[0::2] - 1.0, 2.0 * u[1::2] - 1.0
r2 = x1 * x1 + x2 * x2
ok = (r2 < 1.0) & (r2 != 0.0)
lg, _ = glibc_log.restated_log(r2[ok], table, build)
f = np.sqrt(-2.0 * lg / r2[ok])
g = np.empty(2 * ok.sum())
g[0::2], g[1::2] = f * x2[ok], f * x1[ok]
ref = (np.sqrt(sigma) * g[:need]).reshape(N, H, nu)
_, e = T._device_draw(plan)
got = e.reshape(H, N, nu).transpose(1, 0, 2)
assert np.array_equal(got, ref), float(np.max(np.abs(got - ref)))
print("BUILD_OK", build, int(np.sum(got != np.random.normal(scale=np.sqrt(sigma), size=(N, H, nu)))))
"""


@pytest.mark.parametrize("build", [1, 2])
def test_each_restated_build_of_log_on_the_device(build):
    """The host of a GPU box runs ONE of glibc's two builds of log(); the device carries both
    restatements (csrc/glibc_log.hpp).  Each is forced onto the device in a fresh process
    (AMPC_LEGACY_LOG) and compared bit for bit with numpy's polar method evaluated on the host with
    the CPU restatement of the same build (oracle/glibc_log.c, itself checked against the library's
    machine code for both builds: profiles/r03_glibc_log_validation.log)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AMPC_LEGACY_LOG=str(build))
    out = subprocess.run([sys.executable, "-c", _FORCED_BUILD_SCRIPT.format(root=root, build=build)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "BUILD_OK %d" % build in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_inplace_generator_state_equals_the_get_set_state_path(monkeypatch):
    """The default mode reads and updates numpy's global generator where numpy keeps it
    (autompc_amd._npstate) instead of through get_state() / set_state().  Both routes must give the
    same controls and leave the same generator state, with host draws in between, and numpy's own
    stream must continue from it."""
    from autompc_amd import MLP, MPPI, QuadCost, Task, _npstate
    if _npstate.get() is None:
        pytest.skip("numpy's RandomState layout not recognised on this box")
    g = golden("mppi_c2_pendulum")
    nx = int(g["nx"])
    p = golden_params(nx, 1, g["hidden"], g["activation"], g["mlp_seed"], bool(g["plain_norm"]))
    system = make_system(nx, 1)
    m = MLP(system, n_hidden_layers=len(p["weights"]) - 1, nonlintype=p["activation"],
            **{"hidden_size_%d" % (i + 1): w.shape[0] for i, w in enumerate(p["weights"][:-1])})
    m.weights, m.biases = p["weights"], p["biases"]
    m.xu_means, m.xu_std, m.dy_means, m.dy_std = p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"]
    task = Task(system)
    task.set_cost(QuadCost(system, g["Q"], g["R"], g["F"], goal=g["goal"]))
    task.set_ctrl_bounds([g["bounds"][0]], [g["bounds"][1]])

    def episode(inplace):
        if not inplace:
            monkeypatch.setattr(_npstate, "get", lambda: None)
        np.random.seed(5)
        ctl = MPPI(system, task, m, horizon=int(g["H"]), num_path=int(g["N"]) + 1, sigma=float(g["sigma"]),
                   lmda=float(g["lmda"]))            # odd sample count: a cached Gaussian is carried
        obs = np.array([0.05, -0.02])[:nx] if nx <= 2 else np.full(nx, 0.03)
        constate = np.concatenate([obs, np.zeros(1)])
        us = []
        for k in range(7):
            if k in (2, 5):
                np.random.normal(size=k)             # someone else draws in between
            u, constate = ctl.run(constate, obs)
            us.append(u)
        monkeypatch.undo()
        return np.array(us), np.random.get_state(), np.random.normal(size=5)
    ua, sa, ta = episode(True)
    ub, sb, tb = episode(False)
    np.testing.assert_array_equal(ua, ub)
    np.testing.assert_array_equal(sa[1], sb[1])
    assert sa[2:] == sb[2:]
    np.testing.assert_array_equal(ta, tb)


@pytest.mark.parametrize("N,H,nu,ahead", [(500, 10, 2, None), (37, 3, 1, "2"), (2048, 30, 3, "3"), (300, 12, 1, "5")])
def test_run_ahead_over_many_calls_with_random_interruptions(N, H, nu, ahead, monkeypatch):
    """The raw stream is generated several calls ahead into two alternating buffers (api.cpp:
    legacy_enqueue / legacy_speculate).  60 calls -- through many buffer switches, with host draws
    of random length at random calls (each invalidates the run-ahead), odd counts (cached Gaussian
    carried over) and positions at block boundaries as they come -- must continue numpy's stream
    exactly: normals, key, position, cache after every call."""
    if ahead is not None:
        monkeypatch.setenv("AMPC_LEGACY_AHEAD", ahead)
    sigma = 0.0049
    h, plan = _plan(N, H, sigma, nx=2, nu=nu)
    rng = np.random.default_rng(N + H)
    np.random.seed(99 + N)
    for call in range(60):
        if rng.random() < 0.15:
            np.random.normal(size=int(rng.integers(1, 2000)))        # someone else draws
        st0 = np.random.get_state()
        ref = np.random.normal(scale=np.sqrt(sigma), size=(N, H, nu))
        st_ref = np.random.get_state()
        np.random.set_state(st0)
        st_dev, e = _device_draw(plan)
        got = e.reshape(H, N, nu).transpose(1, 0, 2)
        np.testing.assert_array_equal(st_dev[1], st_ref[1], err_msg="key after call %d" % call)
        assert st_dev[2:] == st_ref[2:], call
        if _exact():
            np.testing.assert_array_equal(got, ref, err_msg="normals of call %d" % call)
        else:
            assert np.max(np.abs(got - ref) / np.spacing(np.abs(ref))) <= 4.0
        np.random.set_state(st_dev)
    plan.close()
    h.close()


def test_concurrent_draws_of_several_plans_continue_their_own_streams():
    """Four plans on four host threads, each with its own generator state, six c3-sized draws each (459 slices
    through at most 256 workgroups: the draw kernel's look-back across launches that run side by side)."""
    import threading
    N, H, nu, sigma = 4096, 30, 6, 0.0049
    made = [_plan(N, H, sigma, nx=2, nu=nu) for _ in range(4)]
    errors = []

    def work(k):
        try:
            h, plan = made[k]
            rs = np.random.RandomState(100 + k)
            state = rs.get_state()
            for call in range(6):
                ref = rs.normal(scale=np.sqrt(sigma), size=(N, H, nu))
                state = plan.legacy_normal(state)
                st_ref = rs.get_state()
                np.testing.assert_array_equal(state[1], st_ref[1], err_msg="plan %d call %d" % (k, call))
                assert tuple(state[2:]) == tuple(st_ref[2:])
                plan.upload(x0=np.zeros((1, 2)), act_seq=np.zeros(plan.sum_hnu))
                plan.solve()
                e = plan.download(act_seq=False, u=False, eps_out=True)[3]
                if _exact():
                    np.testing.assert_array_equal(e.reshape(H, N, nu).transpose(1, 0, 2), ref)
        except Exception as ex:      # noqa: BLE001 -- reported by the main thread
            errors.append((k, repr(ex)[:300]))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for h, plan in made:
        plan.close()
        h.close()
    assert not errors, errors


@pytest.mark.parametrize("shape", [(2, 1, [64, 64], 1024, 30), (17, 6, [256, 256], 512, 12)])
def test_hot_call_is_the_same_with_mapped_io_and_pre_drawn_noise(shape, monkeypatch):
    """MPPI.run() in the parity-graded numpy-stream mode (ampc_mppi_run_legacy): x0 / u through mapped host memory
    with a polled completion word, and the NEXT call's normals drawn behind this call's update from the generator
    state it returns (round 6).  Twelve consecutive control steps -- with somebody else drawing from numpy's
    generator before steps 4 and 9, which must invalidate the pre-drawn noise -- give bit-identical controls,
    states of the global generator and warm starts in all four combinations of the two switches, and equal the
    step-by-step path (upload / legacy_normal / solve / download: return_details=True)."""
    if not _exact():
        pytest.skip("the host's log() is not one of the two glibc builds the library reproduces")
    from autompc_amd import MLP, MPPI, QuadCost, Task
    nx, nu, hidden, N, H = shape
    system = make_system(nx, nu)
    p = omlp.random_params(nx, nu, hidden, "relu", seed=5)

    def controller():
        m = MLP(system, n_hidden_layers=2, nonlintype="relu", hidden_size_1=hidden[0], hidden_size_2=hidden[1])
        m.weights, m.biases = [w.copy() for w in p["weights"]], [b.copy() for b in p["biases"]]
        m.xu_means, m.xu_std, m.dy_means, m.dy_std = p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"]
        task = Task(system)
        task.set_cost(QuadCost(system, np.eye(nx), 0.01 * np.eye(nu), np.eye(nx)))
        task.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
        return MPPI(system, task, m, horizon=H, num_path=N, sigma=0.5, lmda=1.0, noise="numpy_device")

    def episode(details=False):
        np.random.seed(99)
        ctl = controller()
        x = np.random.default_rng(0).uniform(-0.1, 0.1, size=nx)
        state = np.concatenate([x, np.zeros(nu)])
        us, gens = [], []
        for step in range(12):
            if step in (4, 9):
                np.random.normal(size=5 + step)
            u, state = ctl.run(state, x, return_details=details)
            x = ctl.model.pred(x, u)
            us.append(u)
            gens.append(np.random.get_state()[1:4])
        return np.array(us), gens
    runs = {}
    for mapped in ("1", "0"):
        for pre in ("1", "0"):
            monkeypatch.setenv("AMPC_RUN_MAPPED", mapped)
            monkeypatch.setenv("AMPC_LEGACY_PREDRAW", pre)
            runs[mapped, pre] = episode()
    ref_u, ref_g = episode(details=True)
    for key, (us, gens) in runs.items():
        np.testing.assert_array_equal(us, ref_u, err_msg=str(key))
        for a, b in zip(gens, ref_g):
            np.testing.assert_array_equal(a[0], b[0])
            assert a[1:] == b[1:]


def test_an_expired_look_back_wait_is_drawn_again(monkeypatch):
    """the draw kernel's workgroups wait (bounded) for their predecessors' pair counts; with the bound
    forced to ONE poll the first attempt of a many-workgroup draw gives up -- the library draws again from the
    untouched generator state with the full bound: numpy's values and generator state, bit for bit, through the
    stand-alone draw and through the one-call control step (whose solve is repeated with the good noise)."""
    if not _exact():
        pytest.skip("the host's log() is not one of the two glibc builds the library reproduces")
    N, H, nu, sigma = 4096, 30, 6, 0.0049
    monkeypatch.setenv("AMPC_POLAR_SPIN_LIMIT", "1")
    h, plan = _plan(N, H, sigma, nx=2, nu=nu)
    np.random.seed(5)
    for _ in range(3):
        st0 = np.random.get_state()
        ref = np.random.normal(scale=np.sqrt(sigma), size=(N, H, nu))
        st_ref = np.random.get_state()
        np.random.set_state(st0)
        st_dev, e = _device_draw(plan)
        np.testing.assert_array_equal(e.reshape(H, N, nu).transpose(1, 0, 2), ref)
        np.testing.assert_array_equal(st_dev[1], st_ref[1])
        assert st_dev[2:4] == st_ref[2:4]
        np.random.set_state(st_dev)
    assert plan.legacy_redraws() >= 1
    # the hot call: same controls as with the full bound
    from autompc_amd import _npstate
    ls = _npstate.get()
    if ls is not None:
        x0, outs = np.zeros((1, 2)), {}
        for limit in ("1", str(1 << 22)):
            monkeypatch.setenv("AMPC_POLAR_SPIN_LIMIT", limit)
            np.random.seed(11)
            plan.upload(x0=x0, act_seq=np.zeros(plan.sum_hnu))
            outs[limit] = [plan.run_legacy_inplace(x0, None, _npstate.get()).copy() for _ in range(4)]
            outs[limit].append(np.random.get_state()[1].copy())
        for a, b in zip(outs["1"], outs[str(1 << 22)]):
            np.testing.assert_array_equal(a, b)
    plan.close()
    h.close()
