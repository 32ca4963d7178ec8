"""ctypes binding of libautompc_hip.so (C ABI: include/autompc_hip.h).

There is deliberately no CPU fallback: if the HIP library is missing or no
MI355X is visible, everything that computes raises ``AmpcError``.
"""
import atexit
import ctypes
import os
import weakref
from ctypes import POINTER, c_char_p, c_double, c_int, c_uint32, c_uint64, c_void_p

import numpy as np

F64, F32 = 0, 1
ACTIVATIONS = {"relu": 0, "tanh": 1, "sigmoid": 2, "selu": 3}
TERM_REFERENCE, TERM_PER_PARTICLE = 0, 1

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AMPC_LIB") or os.path.join(_HERE, "libautompc_hip.so")


class AmpcError(RuntimeError):
    pass


_lib = None
_dp = POINTER(c_double)
_ip = POINTER(c_int)

# name -> (restype, argtypes); must list every symbol include/autompc_hip.h declares
SIGNATURES = {
    "ampc_last_error": (c_char_p, []),
    "ampc_version": (c_int, []),
    "ampc_device_count": (c_int, []),
    "ampc_create": (c_int, [c_int, c_int, c_void_p, POINTER(c_void_p)]),
    "ampc_destroy": (c_int, [c_void_p]),
    "ampc_synchronize": (c_int, [c_void_p]),
    "ampc_precision": (c_int, [c_void_p]),
    "ampc_set_mlp": (c_int, [c_void_p, c_int, c_int, c_int, _ip, c_int, POINTER(_dp),
                             POINTER(_dp), _dp, _dp, _dp, _dp]),
    "ampc_set_mlp_dev": (c_int, [c_void_p, c_int, c_int, c_int, _ip, c_int, POINTER(c_void_p),
                                 POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p]),
    "ampc_jit_status": (c_int, [c_void_p, c_char_p, c_int]),
    "ampc_handle_set_jit": (c_int, [c_void_p, c_int]),
    "ampc_jit_wait": (c_int, [c_void_p]),
    "ampc_plan_kernel_kind": (c_int, [c_void_p, c_void_p]),
    "ampc_set_linear": (c_int, [c_void_p, c_int, c_int, _dp, _dp]),
    "ampc_mlp_pred_batch": (c_int, [c_void_p, _dp, _dp, _dp, c_int]),
    "ampc_mlp_pred_diff_batch": (c_int, [c_void_p, _dp, _dp, _dp, _dp, _dp, c_int]),
    "ampc_set_sindy": (c_int, [c_void_p, c_int, c_int, c_int, _ip, _ip, _ip, _dp, _dp, c_int, c_double,
                               c_int, c_int, _ip, _ip]),
    "ampc_sindy_pred_batch": (c_int, [c_void_p, _dp, _dp, _dp, c_int]),
    "ampc_sindy_pred_diff_batch": (c_int, [c_void_p, _dp, _dp, _dp, _dp, _dp, c_int]),
    "ampc_set_quad_costs": (c_int, [c_void_p, c_int, c_int, _dp, _dp, _dp, _dp]),
    "ampc_set_indicator_costs": (c_int, [c_void_p, c_int, _ip, _dp]),
    "ampc_set_affine_quad_costs": (c_int, [c_void_p, c_int, c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "ampc_set_ctrl_bounds": (c_int, [c_void_p, _dp, _dp]),
    "ampc_mppi_plan_create": (c_int, [c_void_p, c_int, _ip, _ip, _dp, _dp, _ip, c_int,
                                      POINTER(c_void_p)]),
    "ampc_mppi_plan_destroy": (c_int, [c_void_p]),
    "ampc_mppi_upload": (c_int, [c_void_p, _dp, _dp, _dp]),
    "ampc_mppi_generate_eps": (c_int, [c_void_p, c_uint64, c_uint64]),
    "ampc_mppi_plan_legacy_redraws": (c_int, [c_void_p, POINTER(ctypes.c_longlong)]),
    "ampc_mppi_plan_set_noise_ids": (c_int, [c_void_p, POINTER(c_uint32)]),
    "ampc_set_mt_jump_table": (c_int, [POINTER(c_uint32), c_int, c_int]),
    "ampc_legacy_log_mode": (c_int, []),
    "ampc_mppi_legacy_normal": (c_int, [c_void_p, POINTER(c_uint32), c_int, c_int, c_double,
                                        POINTER(c_uint32), _ip, _ip, _dp]),
    "ampc_mppi_plan_set_geometry": (c_int, [c_void_p, c_int, c_int]),
    "ampc_mppi_plan_set_step_offset": (c_int, [c_void_p, c_uint64]),
    "ampc_mppi_plan_set_state_lift": (c_int, [c_void_p, c_int, _ip, _dp]),
    "ampc_mppi_solve": (c_int, [c_void_p]),
    "ampc_mppi_download": (c_int, [c_void_p, _dp, _dp, _dp, _dp]),
    "ampc_mppi_set_x0_dev": (c_int, [c_void_p, c_void_p]),
    "ampc_mppi_run": (c_int, [c_void_p, _dp, _dp, c_int, c_uint64, c_uint64, _dp]),
    "ampc_mppi_run_legacy": (c_int, [c_void_p, _dp, _dp, POINTER(c_uint32), c_int, c_int, c_double,
                                     POINTER(c_uint32), _ip, _ip, _dp, _dp]),
    "ampc_mppi_plan_info": (c_int, [c_void_p, _ip, _ip, _dp, _dp]),
    "ampc_mppi_plan_set_outputs": (c_int, [c_void_p, c_int]),
    "ampc_mppi_plan_set_timing": (c_int, [c_void_p, c_int]),
    "ampc_mppi_plan_timing": (c_int, [c_void_p, _dp, _dp, _ip]),
    "ampc_mppi_closed_loop": (c_int, [c_void_p, c_void_p, _dp, c_int, c_uint64, _dp, _dp, _dp]),
    "ampc_score_trajectories": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, _dp, _dp, c_int,
                                        _ip, _dp, _dp]),
    "ampc_mppi_closed_loop_scored": (c_int, [c_void_p, c_void_p, _dp, c_int, c_uint64, _dp, c_int,
                                             _ip, _dp, _dp, _dp, _dp]),
    "ampc_ilqr_plan_create": (c_int, [c_void_p, c_int, c_int, c_double, _ip, c_int,
                                      POINTER(c_void_p)]),
    "ampc_ilqr_plan_destroy": (c_int, [c_void_p]),
    "ampc_ilqr_plan_set_constants": (c_int, [c_void_p, c_double, c_int, c_double, c_double]),
    "ampc_ilqr_plan_set_terminal_goal": (c_int, [c_void_p, c_int]),
    "ampc_ilqr_plan_set_timing": (c_int, [c_void_p, c_int]),
    "ampc_ilqr_plan_timing": (c_int, [c_void_p, _dp, _ip]),
    "ampc_ilqr_plan_stats": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong)]),
    "ampc_ilqr_solve": (c_int, [c_void_p, _dp, _dp, c_int, _dp, _dp, _dp, _dp, _ip, _ip, _ip, _dp]),
    "ampc_ilqr_closed_loop": (c_int, [c_void_p, c_void_p, c_int, _dp, _ip, c_int, c_int, _dp, _dp, _ip, _ip,
                                      POINTER(ctypes.c_longlong)]),
    "ampc_ilqr_solve_queue": (c_int, [c_void_p, c_int, _dp, _dp, _ip, c_int, _dp, _dp, _dp, _dp, _ip, _ip, _ip,
                                      _dp]),
    "ampc_ilqr_solve_queue_var": (c_int, [c_void_p, c_int, _dp, _dp, _ip, _ip, _ip, c_int, _dp, _dp, _dp, _dp, _ip, _ip,
                                          _ip, _dp]),
    "ampc_ilqr_closed_loop_var": (c_int, [c_void_p, c_void_p, c_int, _dp, _ip, _ip, _ip, c_int, c_int, _dp, _dp, _ip,
                                          _ip, POINTER(ctypes.c_longlong)]),
    "ampc_ilqr_plan_set_models": (c_int, [c_void_p, c_int, POINTER(c_void_p)]),
    "ampc_mppi_plan_set_models": (c_int, [c_void_p, c_int, POINTER(c_void_p), _ip]),
}


def load():
    """Load the shared library (once) and attach prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AmpcError("libautompc_hip.so is not built (%s). Run `python -c 'import "
                        "__graft_entry__ as g; g.build()'` or `make -C autompc_amd/csrc`."
                        % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def legacy_log_mode():
    """ampc_legacy_log_mode(): 1 / 2 when the library reproduces the host C library's log() bit for
    bit (glibc's FMA / non-FMA build) -- the device-generated numpy legacy stream is then numpy's
    own, normals included -- 0 when it does not."""
    return int(load().ampc_legacy_log_mode())


def check(rc):
    if rc != 0:
        msg = load().ampc_last_error()
        raise AmpcError(msg.decode() if msg else "libautompc_hip error %d" % rc)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def dptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def iptr(a):
    return None if a is None else a.ctypes.data_as(_ip)


# Device objects must be destroyed before the HIP runtime's own static teardown: close every live
# plan, then every live handle, from an atexit hook (runs before interpreter/module teardown).
_live_plans = weakref.WeakSet()
_live_handles = weakref.WeakSet()


@atexit.register
def _shutdown():
    for obj in list(_live_plans) + list(_live_handles):
        try:
            obj.close()
        except Exception:
            pass


class Handle:
    """One device context: model + cost blocks + bounds on one MI355X / one stream."""

    def __init__(self, device=0, precision="f64", stream=None, jit=True):
        """jit=False: never start (or wait for) the run-time build of shape-specialised kernels for what this
        handle holds (ampc_handle_set_jit) -- models that live for one evaluation."""
        lib = load()
        if lib.ampc_device_count() <= 0:
            raise AmpcError("no HIP device visible: the MI355X path cannot run here "
                            "(there is no CPU fallback by design)")
        self.lib = lib
        self.precision = {"f64": F64, "f32": F32}[precision]
        self.precision_name = precision
        self._h = c_void_p()
        check(lib.ampc_create(int(device), self.precision, c_void_p(stream) if stream else None,
                              ctypes.byref(self._h)))
        self.device = int(device)
        if not jit:
            check(lib.ampc_handle_set_jit(self._h, 0))
        self.nx = self.nu = None
        self._plans = weakref.WeakSet()      # plans hold raw pointers into this handle
        _live_handles.add(self)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            # garbage collection may finalise a handle before the plans built on it: destroy
            # those first, they dereference the handle
            for plan in list(getattr(self, "_plans", ())):
                plan.close()
            self.lib.ampc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(self.lib.ampc_synchronize(self._h))

    # -- model ---------------------------------------------------------------
    def set_mlp(self, nx, nu, weights, biases, activation, xu_means, xu_std, dy_means, dy_std):
        n_hidden = len(weights) - 1
        Ws = [as_f64(w) for w in weights]
        bs = [as_f64(b) for b in biases]
        hidden = np.array([w.shape[0] for w in Ws[:-1]], dtype=np.int32)
        sizes = [nx + nu] + [int(x) for x in hidden] + [nx]
        for l, w in enumerate(Ws):
            if w.shape != (sizes[l + 1], sizes[l]) or bs[l].shape != (sizes[l + 1],):
                raise ValueError("layer %d has shape %r, expected %r" % (l, w.shape,
                                                                          (sizes[l + 1], sizes[l])))
        wp = (_dp * len(Ws))(*[dptr(w) for w in Ws])
        bp = (_dp * len(bs))(*[dptr(b) for b in bs])
        norm = [as_f64(v) for v in (xu_means, xu_std, dy_means, dy_std)]
        if norm[0].shape != (nx + nu,) or norm[1].shape != (nx + nu,) \
                or norm[2].shape != (nx,) or norm[3].shape != (nx,):
            raise ValueError("normaliser shapes do not match (nx+nu, nx+nu, nx, nx)")
        check(self.lib.ampc_set_mlp(self._h, nx, nu, n_hidden, iptr(hidden), ACTIVATIONS[activation],
                                    wp, bp, dptr(norm[0]), dptr(norm[1]), dptr(norm[2]),
                                    dptr(norm[3])))
        self.nx, self.nu = nx, nu
        self._sindy = False

    def set_mlp_dev(self, nx, nu, hidden, weight_ptrs, bias_ptrs, activation, norm_ptrs):
        """Stage an MLP whose float64 parameters already live in this device's memory (e.g. the tensors a
        PyTorch-ROCm fit produced): `weight_ptrs[l]` / `bias_ptrs[l]` are device addresses of contiguous
        [out_l][in_l] / [out_l] arrays, `norm_ptrs` those of xu_means, xu_std, dy_means, dy_std.  Folding
        and packing run on the device (csrc/api_model.cpp); the caller's arrays are only read, and must
        be complete (their producing stream synchronised) when this is called."""
        hidden = np.array([int(x) for x in hidden], dtype=np.int32)
        if len(weight_ptrs) != len(hidden) + 1 or len(bias_ptrs) != len(hidden) + 1 or len(norm_ptrs) != 4:
            raise ValueError("one weight and one bias pointer per layer (hidden + output), four normaliser pointers")
        wp = (c_void_p * len(weight_ptrs))(*[c_void_p(int(p)) for p in weight_ptrs])
        bp = (c_void_p * len(bias_ptrs))(*[c_void_p(int(p)) for p in bias_ptrs])
        check(self.lib.ampc_set_mlp_dev(self._h, int(nx), int(nu), len(hidden), iptr(hidden), ACTIVATIONS[activation],
                                        wp, bp, *[c_void_p(int(p)) for p in norm_ptrs]))
        self.nx, self.nu = int(nx), int(nu)
        self._sindy = False

    def jit_status(self):
        """(state, message): 0 no run-time compiled kernels involved, 1 the shape plugin of the staged
        model is being compiled, 2 ready, -1 failed."""
        buf = ctypes.create_string_buffer(512)
        st = self.lib.ampc_jit_status(self._h, buf, 512)
        msg = buf.value.decode()
        if st == -1 and not getattr(self, "_jit_warned", False):
            # say it once: the controller keeps working, on the run-time-shape kernels (1.1-1.9x slower)
            import warnings
            self._jit_warned = True
            warnings.warn("autompc_amd: the kernels specialised for this model's shape could not be compiled at run "
                          "time (%s); plans on this model run the run-time-shape kernels, which are 1.1-1.9x slower. "
                          "A box without hipcc can ship a cache built elsewhere ($AMPC_JIT_CACHE)." % (msg or "no log"),
                          RuntimeWarning, stacklevel=3)
        return st, msg

    def jit_wait(self):
        """Block until the kernels specialised for the staged model's shape are compiled; plans
        created afterwards use them."""
        check(self.lib.ampc_jit_wait(self._h))

    def set_linear(self, A, B):
        """x' = A x + B u (ARX / Koopman prediction, arx.py:151-154, koopman.py:170-173)."""
        A, B = as_f64(A), as_f64(B)
        nx = A.shape[0]
        if A.ndim != 2 or A.shape != (nx, nx) or B.ndim != 2 or B.shape[0] != nx:
            raise ValueError("A must be [nx, nx] and B [nx, nu]")
        check(self.lib.ampc_set_linear(self._h, nx, B.shape[1], dptr(A), dptr(B)))
        self.nx, self.nu = nx, B.shape[1]
        self._sindy = False

    def set_sindy(self, nx, nu, kind, arg0, arg1, param, xi, continuous, dt, strict_reference=True,
                  pair_var=None, pair_exp=None):
        """pair_var / pair_exp: the (variable, exponent) pairs monomial features (kind 6) index
        with arg0 (first pair) and arg1 (number of pairs)."""
        kind = np.ascontiguousarray(kind, dtype=np.int32)
        pair_var = np.ascontiguousarray(pair_var if pair_var is not None else [], dtype=np.int32)
        pair_exp = np.ascontiguousarray(pair_exp if pair_exp is not None else [], dtype=np.int32)
        if pair_var.shape != pair_exp.shape or pair_var.ndim != 1:
            raise ValueError("pair_var and pair_exp must be 1-d arrays of equal length")
        arg0 = np.ascontiguousarray(arg0, dtype=np.int32)
        arg1 = np.ascontiguousarray(arg1, dtype=np.int32)
        param, xi = as_f64(param), as_f64(xi)
        nf = kind.shape[0]
        if xi.shape != (nx, nf) or arg0.shape != (nf,) or arg1.shape != (nf,) or param.shape != (nf,):
            raise ValueError("SINDy descriptor shapes are inconsistent")
        check(self.lib.ampc_set_sindy(self._h, nx, nu, nf, iptr(kind), iptr(arg0), iptr(arg1),
                                      dptr(param), dptr(xi), int(bool(continuous)),
                                      float(dt or 0.0), int(bool(strict_reference)),
                                      int(pair_var.shape[0]), iptr(pair_var) if pair_var.size else None,
                                      iptr(pair_exp) if pair_exp.size else None))
        self.nx, self.nu = nx, nu
        self._sindy = True

    def pred_batch(self, states, ctrls):
        states, ctrls = as_f64(states), as_f64(ctrls)
        n = states.shape[0]
        if states.shape != (n, self.nx) or ctrls.shape != (n, self.nu):
            raise ValueError("pred_batch: states %r / ctrls %r do not match (n,%d)/(n,%d)"
                             % (states.shape, ctrls.shape, self.nx, self.nu))
        out = np.empty((n, self.nx))
        fn = self.lib.ampc_sindy_pred_batch if getattr(self, "_sindy", False) \
            else self.lib.ampc_mlp_pred_batch
        check(fn(self._h, dptr(states), dptr(ctrls), dptr(out), n))
        return out

    def pred_diff_batch(self, states, ctrls):
        states, ctrls = as_f64(states), as_f64(ctrls)
        n = states.shape[0]
        if states.shape != (n, self.nx) or ctrls.shape != (n, self.nu):
            raise ValueError("pred_diff_batch: bad shapes %r %r" % (states.shape, ctrls.shape))
        out = np.empty((n, self.nx))
        jx = np.empty((n, self.nx, self.nx))
        ju = np.empty((n, self.nx, self.nu))
        fn = self.lib.ampc_sindy_pred_diff_batch if getattr(self, "_sindy", False) \
            else self.lib.ampc_mlp_pred_diff_batch
        check(fn(self._h, dptr(states), dptr(ctrls), dptr(out), dptr(jx), dptr(ju), n))
        return out, jx, ju

    # -- cost / bounds ----------------------------------------------------------
    def set_quad_costs(self, Q, R, F, goal):
        """Q [C,no,no], R [C,nu,nu], F [C,no,no], goal [C,no] (or a single block without C)."""
        Q, R, F, goal = as_f64(Q), as_f64(R), as_f64(F), as_f64(goal)
        if Q.ndim == 2:
            Q, R, F, goal = Q[None], R[None], F[None], goal[None]
        C, no = Q.shape[0], Q.shape[1]
        if Q.shape != (C, no, no) or F.shape != (C, no, no) or R.shape != (C, self.nu, self.nu) \
                or goal.shape != (C, no):
            raise ValueError("cost block shapes are inconsistent")
        check(self.lib.ampc_set_quad_costs(self._h, C, no, dptr(Q), dptr(R), dptr(F), dptr(goal)))
        self.obs_dim = no
        self.n_costs = C

    def set_cost_blocks(self, Q, R, F, goal, lin=None, lin_term=None, consts=None):
        """Affine-quadratic cost blocks (ampc_set_affine_quad_costs; autompc_amd.costs.blocks): the
        arrays of set_quad_costs plus lin [C,no], lin_term [C,no], consts [C,2] (None = zeros)."""
        Q, R, F, goal = as_f64(Q), as_f64(R), as_f64(F), as_f64(goal)
        if Q.ndim == 2:
            Q, R, F, goal = Q[None], R[None], F[None], goal[None]
            lin, lin_term, consts = (None if a is None else as_f64(a)[None] for a in (lin, lin_term, consts))
        C, no = Q.shape[0], Q.shape[1]
        if Q.shape != (C, no, no) or F.shape != (C, no, no) or R.shape != (C, self.nu, self.nu) \
                or goal.shape != (C, no):
            raise ValueError("cost block shapes are inconsistent")
        extra = []
        for a, shape in ((lin, (C, no)), (lin_term, (C, no)), (consts, (C, 2))):
            if a is not None:
                a = as_f64(a)
                if a.shape != shape:
                    raise ValueError("cost block shapes are inconsistent")
            extra.append(a)
        check(self.lib.ampc_set_affine_quad_costs(self._h, C, no, dptr(Q), dptr(R), dptr(F), dptr(goal),
                                                  *(dptr(a) if a is not None else None for a in extra)))
        self.obs_dim = no
        self.n_costs = C

    def set_indicator_costs(self, terms=None):
        """Threshold / box terms of an MPPI controller's cost (ampc_set_indicator_costs): `terms` =
        (kinds int32[n], params f64[...]) as autompc_amd.costs.cost_terms lays them out, kinds 1 / 2 only;
        None or empty removes them.  Call after set_quad_costs / set_cost_blocks."""
        if terms is None or len(terms[0]) == 0:
            check(self.lib.ampc_set_indicator_costs(self._h, 0, None, None))
            self.n_ind = 0
            return
        kinds = np.ascontiguousarray(terms[0], dtype=np.int32)
        params = as_f64(terms[1])
        check(self.lib.ampc_set_indicator_costs(self._h, int(kinds.shape[0]), iptr(kinds), dptr(params)))
        self.n_ind = int(kinds.shape[0])

    def set_ctrl_bounds(self, lo, hi):
        lo, hi = as_f64(lo), as_f64(hi)
        check(self.lib.ampc_set_ctrl_bounds(self._h, dptr(lo), dptr(hi)))

    def score_trajectories(self, terms, obs, ctrls, obs_dim=None):
        """Cost.__call__ (cost.py:27-41) of a batch of trajectories obs [B,T,ns], ctrls [B,T,nu]
        under the cost given as `terms` = (kinds int32[n], params f64[...]) -- see
        autompc_amd.costs.cost_terms.  Returns scores [B]."""
        kinds, params = terms
        obs, ctrls = as_f64(obs), as_f64(ctrls)
        if obs.ndim != 3 or ctrls.ndim != 3 or obs.shape[:2] != ctrls.shape[:2]:
            raise ValueError("obs [B,T,ns] and ctrls [B,T,nu] expected")
        B, T, ns = obs.shape
        no = ns if obs_dim is None else int(obs_dim)
        kinds = np.ascontiguousarray(kinds, dtype=np.int32)
        params = as_f64(params)
        out = np.empty(B)
        check(self.lib.ampc_score_trajectories(
            self._h, B, T, ns, no, ctrls.shape[2], dptr(obs), dptr(ctrls), len(kinds),
            kinds.ctypes.data_as(_ip), dptr(params), dptr(out)))
        return out


class MppiPlan:
    """Device buffers + launch geometry for a batch of independent MPPI problems."""

    def __init__(self, handle, num_path, horizon, sigma, lmda, cost_index=None,
                 term_mode=TERM_REFERENCE):
        self.handle = handle
        self.lib = handle.lib
        self.N = np.atleast_1d(np.asarray(num_path, dtype=np.int32)).copy()
        self.B = self.N.shape[0]
        self.H = np.broadcast_to(np.asarray(horizon, dtype=np.int32), (self.B,)).copy()
        self.sigma = np.broadcast_to(as_f64(sigma), (self.B,)).copy()
        self.lmda = np.broadcast_to(as_f64(lmda), (self.B,)).copy()
        ci = None if cost_index is None else \
            np.broadcast_to(np.asarray(cost_index, dtype=np.int32), (self.B,)).copy()
        self._p = c_void_p()
        check(self.lib.ampc_mppi_plan_create(handle._h, self.B, iptr(self.N), iptr(self.H),
                                             dptr(self.sigma), dptr(self.lmda), iptr(ci),
                                             int(term_mode), ctypes.byref(self._p)))
        _live_plans.add(self)
        handle._plans.add(self)
        nu = handle.nu
        self.sum_hnu = int(np.sum(self.H.astype(np.int64) * nu))
        self.sum_n = int(np.sum(self.N.astype(np.int64)))
        self.sum_nhnu = int(np.sum(self.N.astype(np.int64) * self.H * nu))

    def close(self):
        if getattr(self, "_p", None) is not None and self._p:
            self.lib.ampc_mppi_plan_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _flat(self, a, size, name):
        if a is None:
            return None
        a = as_f64(a).reshape(-1)
        if a.size != size:
            raise ValueError("%s has %d elements, expected %d" % (name, a.size, size))
        return a

    def upload(self, x0=None, act_seq=None, eps=None):
        x0 = self._flat(x0, self.B * self.handle.nx, "x0")
        act_seq = self._flat(act_seq, self.sum_hnu, "act_seq")
        eps = self._flat(eps, self.sum_nhnu, "eps")
        check(self.lib.ampc_mppi_upload(self._p, dptr(x0), dptr(act_seq), dptr(eps)))

    def generate_eps(self, seed, stream=0):
        check(self.lib.ampc_mppi_generate_eps(self._p, int(seed), int(stream)))

    _jump_table_set = False

    @classmethod
    def _install_jump_table(cls, lib):
        """MT19937 jump polynomials for the block-parallel stream generator (once per process);
        without the data file the library generates the stream sequentially."""
        if cls._jump_table_set:
            return
        cls._jump_table_set = True
        path = os.path.join(_HERE, "data", "mt19937_jump.npz")
        if os.path.exists(path):
            d = np.load(path)
            for j in d["jumps"]:           # one table per segment length
                polys = np.ascontiguousarray(d["polys_%d" % int(j)], dtype=np.uint32)
                check(lib.ampc_set_mt_jump_table(polys.ctypes.data_as(POINTER(c_uint32)), polys.shape[0], int(j)))

    def legacy_normal(self, state):
        """Fill the noise buffer with numpy's legacy normal draw for the generator state `state`
        (the tuple np.random.get_state() returns); returns the state to install afterwards."""
        self._install_jump_table(self.lib)
        name, key, pos, has_gauss, cached = state
        if name != "MT19937":
            raise ValueError("numpy's legacy global generator (MT19937) expected")
        key = np.ascontiguousarray(key, dtype=np.uint32)
        key_out = np.empty(624, dtype=np.uint32)
        pos_out, hg_out, cached_out = c_int(), c_int(), c_double()
        u32p = POINTER(c_uint32)
        check(self.lib.ampc_mppi_legacy_normal(self._p, key.ctypes.data_as(u32p), int(pos), int(has_gauss),
                                               float(cached), key_out.ctypes.data_as(u32p),
                                               ctypes.byref(pos_out), ctypes.byref(hg_out),
                                               ctypes.byref(cached_out)))
        return ("MT19937", key_out, pos_out.value, hg_out.value, cached_out.value)

    def legacy_normal_inplace(self, ls):
        """legacy_normal() on numpy's global generator where it lives (ls: _npstate.LegacyState):
        the library reads the MT19937 key from numpy's memory and writes the state after the draw
        back into it -- no get_state() / set_state() conversions (~100 us of a 450 us call)."""
        self._install_jump_table(self.lib)
        pos_out, hg_out, cached_out = c_int(), c_int(), c_double()
        with ls.lock:
            check(self.lib.ampc_mppi_legacy_normal(self._p, ls.key_ptr, int(ls.key[624]), ls.has_gauss.value,
                                                   ls.gauss.value, ls.key_ptr, ctypes.byref(pos_out),
                                                   ctypes.byref(hg_out), ctypes.byref(cached_out)))
            ls.key[624] = pos_out.value
            ls.has_gauss.value, ls.gauss.value = hg_out.value, cached_out.value

    def set_geometry(self, tile_rows=0, horizon_cap=0):
        """Fix the rollout tile height (0 = automatic, 16/32/64) and the horizon the LDS layout is
        sized for, so results do not depend on what else shares the plan.  Call before upload()."""
        check(self.lib.ampc_mppi_plan_set_geometry(self._p, int(tile_rows), int(horizon_cap)))

    def set_step_offset(self, first_step):
        """Index of the first control step of the next closed loop (its Philox stream key), for
        episodes run in segments."""
        check(self.lib.ampc_mppi_plan_set_step_offset(self._p, int(first_step)))

    def set_noise_ids(self, ids):
        """ids [B]: the key of every problem's device noise stream (default: its index in the
        plan).  With a candidate's global index here its noise is independent of the sharding."""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        if ids.shape != (self.B,):
            raise ValueError("one noise id per problem expected")
        check(self.lib.ampc_mppi_plan_set_noise_ids(self._p, ids.ctypes.data_as(POINTER(c_uint32))))

    def set_models(self, handles, model_index):
        """Controller models per problem (ampc_mppi_plan_set_models): `handles` hold MLPs of the plan's shape,
        model_index [B] names each problem's.  None / empty handles: back to the plan handle's own model."""
        hs = list(handles or [])
        arr = (c_void_p * max(len(hs), 1))(*[h._h for h in hs])
        mi = None if not hs else np.ascontiguousarray(np.broadcast_to(np.asarray(model_index, dtype=np.int32), (self.B,)))
        check(self.lib.ampc_mppi_plan_set_models(self._p, len(hs), arr, iptr(mi)))
        self._models = hs

    def solve(self):
        check(self.lib.ampc_mppi_solve(self._p))

    def download(self, act_seq=True, u=True, costs=False, eps_out=False):
        nu = self.handle.nu
        a = np.empty(self.sum_hnu) if act_seq else None
        uu = np.empty((self.B, nu)) if u else None
        c = np.empty(self.sum_n) if costs else None
        e = np.empty(self.sum_nhnu) if eps_out else None
        check(self.lib.ampc_mppi_download(self._p, dptr(a), dptr(uu), dptr(c), dptr(e)))
        return a, uu, c, e

    def run(self, x0, act_seq=None, philox=None):
        """One control step in a single call: x0 (and optionally a new warm start) in, noise (the
        buffer as it is, or fresh Philox noise when philox=(seed, stream)), solve, controls
        [B, nu] out."""
        nx, nu = self.handle.nx, self.handle.nu
        x0 = self._flat(x0, self.B * nx, "x0")
        act_seq = self._flat(act_seq, self.sum_hnu, "act_seq")
        u = np.empty((self.B, nu))
        seed, stream = philox if philox is not None else (0, 0)
        check(self.lib.ampc_mppi_run(self._p, dptr(x0), dptr(act_seq), 0 if philox is None else 1,
                                     int(seed), int(stream), dptr(u)))
        return u

    def legacy_redraws(self):
        """Draws repeated because a bounded wait inside the draw kernel expired (ampc_mppi_plan_legacy_redraws)."""
        n = ctypes.c_longlong()
        check(self.lib.ampc_mppi_plan_legacy_redraws(self._p, ctypes.byref(n)))
        return int(n.value)

    def run_legacy_inplace(self, x0, act_seq, ls):
        """run() with the reference's own noise: numpy's legacy draw from the global generator
        where it lives (ls: _npstate.LegacyState; see legacy_normal_inplace) and the solve, in one
        library call with one synchronisation (ampc_mppi_run_legacy)."""
        self._install_jump_table(self.lib)
        nx, nu = self.handle.nx, self.handle.nu
        x0 = self._flat(x0, self.B * nx, "x0")
        act_seq = self._flat(act_seq, self.sum_hnu, "act_seq")
        u = np.empty((self.B, nu))
        pos_out, hg_out, cached_out = c_int(), c_int(), c_double()
        with ls.lock:
            check(self.lib.ampc_mppi_run_legacy(self._p, dptr(x0), dptr(act_seq), ls.key_ptr, int(ls.key[624]),
                                                ls.has_gauss.value, ls.gauss.value, ls.key_ptr,
                                                ctypes.byref(pos_out), ctypes.byref(hg_out),
                                                ctypes.byref(cached_out), dptr(u)))
            ls.key[624] = pos_out.value
            ls.has_gauss.value, ls.gauss.value = hg_out.value, cached_out.value
        return u

    def set_x0_dev(self, ptr):
        check(self.lib.ampc_mppi_set_x0_dev(self._p, c_void_p(ptr)))

    def set_state_lift(self, kinds, params):
        """The controller model rebuilds its state from every observation (Koopman): x0 = the basis
        functions (kind 0 identity, 1 power, 2 sin, 3 cos; parameter) applied to the observation.  The
        closed loop then carries the simulation model's state: init_obs and the recorded rows have the
        SURROGATE's state width."""
        kinds = np.ascontiguousarray(kinds, dtype=np.int32)
        params = as_f64(params)
        check(self.lib.ampc_mppi_plan_set_state_lift(self._p, len(kinds), iptr(kinds), dptr(params)))
        self._lift = len(kinds) > 0

    def _loop_width(self, surrogate):
        if getattr(self, "_lift", False):
            return (surrogate if surrogate is not None else self.handle).nx
        return self.handle.nx

    def closed_loop(self, init_obs, n_steps, seed=0, eps_all=None, surrogate=None):
        """simulate() for every problem of the plan, device resident.  Returns
        (traj_obs [B, n_steps+1, nx], traj_ctrls [B, n_steps+1, nu])."""
        nx, nu = self._loop_width(surrogate), self.handle.nu
        init_obs = self._flat(init_obs, self.B * nx, "init_obs")
        eps_all = self._flat(eps_all, n_steps * self.sum_nhnu, "eps_all")
        obs = np.empty((self.B, n_steps + 1, nx))
        ctl = np.empty((self.B, n_steps + 1, nu))
        check(self.lib.ampc_mppi_closed_loop(self._p, surrogate._h if surrogate is not None else None,
                                             dptr(init_obs), int(n_steps), int(seed), dptr(eps_all),
                                             dptr(obs), dptr(ctl)))
        return obs, ctl

    def closed_loop_scored(self, init_obs, n_steps, terms, seed=0, eps_all=None, surrogate=None,
                           return_trajectories=False):
        """closed_loop + cost(traj) on the device (eval_cfg's simulate + score,
        pipeline_tuner.py:222-233); only the B scores come back unless return_trajectories."""
        nx, nu = self._loop_width(surrogate), self.handle.nu
        init_obs = self._flat(init_obs, self.B * nx, "init_obs")
        eps_all = self._flat(eps_all, n_steps * self.sum_nhnu, "eps_all")
        kinds = np.ascontiguousarray(terms[0], dtype=np.int32)
        params = as_f64(terms[1])
        scores = np.empty(self.B)
        obs = np.empty((self.B, n_steps + 1, nx)) if return_trajectories else None
        ctl = np.empty((self.B, n_steps + 1, nu)) if return_trajectories else None
        check(self.lib.ampc_mppi_closed_loop_scored(
            self._p, surrogate._h if surrogate is not None else None, dptr(init_obs), int(n_steps),
            int(seed), dptr(eps_all), len(kinds), kinds.ctypes.data_as(_ip), dptr(params),
            dptr(scores), dptr(obs), dptr(ctl)))
        return (scores, obs, ctl) if return_trajectories else scores

    def kernel_kind(self):
        """0 run-time-shape kernels, 1 a shape registered at build time, 2 a run-time compiled plugin."""
        return int(self.lib.ampc_plan_kernel_kind(self._p, None))

    def set_outputs(self, keep_eps_out=True):
        check(self.lib.ampc_mppi_plan_set_outputs(self._p, int(bool(keep_eps_out))))

    def set_timing(self, enable=True, every=1):
        """Bracket the kernels of every `every`-th solve with HIP events (timing())."""
        check(self.lib.ampc_mppi_plan_set_timing(self._p, max(1, int(every)) if enable else 0))

    def timing(self):
        r, u, n = c_double(), c_double(), c_int()
        check(self.lib.ampc_mppi_plan_timing(self._p, ctypes.byref(r), ctypes.byref(u),
                                             ctypes.byref(n)))
        return {"rollout_ms": r.value, "update_ms": u.value, "count": n.value}

    def info(self):
        wg, spw = c_int(), c_int()
        fl, by = c_double(), c_double()
        check(self.lib.ampc_mppi_plan_info(self._p, ctypes.byref(wg), ctypes.byref(spw),
                                           ctypes.byref(fl), ctypes.byref(by)))
        return {"workgroups": wg.value, "samples_per_wg": spw.value, "flops": fl.value,
                "bytes": by.value}


class IlqrPlan:
    """Device buffers for a batch of B independent iLQR problems of horizon H."""

    def __init__(self, handle, B, horizon, dt, cost_index=None, clip_to_bounds=False,
                 terminal_goal=False):
        self.handle = handle
        self.lib = handle.lib
        self.B, self.H = int(B), int(horizon)
        ci = None if cost_index is None else \
            np.broadcast_to(np.asarray(cost_index, dtype=np.int32), (self.B,)).copy()
        self._p = c_void_p()
        check(self.lib.ampc_ilqr_plan_create(handle._h, self.B, self.H, float(dt), iptr(ci),
                                             int(bool(clip_to_bounds)), ctypes.byref(self._p)))
        _live_plans.add(self)
        handle._plans.add(self)
        if terminal_goal:      # QuadCost(strict_reference=False): terminal gradient about the goal
            check(self.lib.ampc_ilqr_plan_set_terminal_goal(self._p, 1))

    def close(self):
        if getattr(self, "_p", None) is not None and self._p:
            self.lib.ampc_ilqr_plan_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_constants(self, u_threshold=1e-3, ls_max_iter=10, ls_discount=0.2, ls_cost_threshold=0.3):
        """compute_ilqr_default's keyword constants (ilqr.py:100-101) for the solves that follow."""
        check(self.lib.ampc_ilqr_plan_set_constants(self._p, float(u_threshold), int(ls_max_iter), float(ls_discount),
                                                    float(ls_cost_threshold)))

    def set_timing(self, enable=True, every=1):
        """Bracket the kernels of an iteration with HIP events (timing()); every > 1: only every n-th iteration of a
        queue (solve_queue)."""
        check(self.lib.ampc_ilqr_plan_set_timing(self._p, max(1, int(every)) if enable else 0))

    def timing(self):
        ms = np.zeros(4)
        n = c_int()
        check(self.lib.ampc_ilqr_plan_timing(self._p, dptr(ms), ctypes.byref(n)))
        return {"riccati_ms": ms[0], "iter_ms": ms[1], "forward_ms": ms[2], "jacobian_ms": ms[3],
                "launches": n.value}

    def kernel_kind(self):
        return int(self.lib.ampc_plan_kernel_kind(None, self._p))

    def stats(self):
        """Work of the last solve: iterations launched and line-search candidate rows rolled out
        (summed over the plan's problems)."""
        it, rows = ctypes.c_longlong(), ctypes.c_longlong()
        check(self.lib.ampc_ilqr_plan_stats(self._p, ctypes.byref(it), ctypes.byref(rows)))
        return {"iterations": it.value, "candidate_rows": rows.value}

    def solve(self, x0, uguess, max_iter=50):
        nx, nu, B, H = self.handle.nx, self.handle.nu, self.B, self.H
        x0 = as_f64(x0).reshape(B, nx)
        uguess = as_f64(uguess).reshape(B, H, nu)
        out = {"states": np.empty((B, H + 1, nx)), "ctrls": np.empty((B, H, nu)),
               "Ks": np.empty((B, H, nu, nx)), "ks": np.empty((B, H, nu)),
               "converged": np.zeros(B, dtype=np.int32), "iters": np.zeros(B, dtype=np.int32),
               "status": np.zeros(B, dtype=np.int32), "objective": np.empty(B)}
        check(self.lib.ampc_ilqr_solve(self._p, dptr(x0), dptr(uguess), int(max_iter),
                                       dptr(out["states"]), dptr(out["ctrls"]), dptr(out["Ks"]),
                                       dptr(out["ks"]), iptr(out["converged"]), iptr(out["iters"]),
                                       iptr(out["status"]), dptr(out["objective"])))
        return out

    def _horizons(self, horizon, n):
        if horizon is None:
            return None
        hz = np.ascontiguousarray(np.broadcast_to(np.asarray(horizon, dtype=np.int32), (n,)))
        if hz.min() < 1 or hz.max() > self.H:
            raise ValueError("horizons must lie in [1, %d] (the plan's horizon)" % self.H)
        return hz

    def set_models(self, handles):
        """Controller models per problem (ampc_ilqr_plan_set_models): `handles` hold MLPs of the plan's shape;
        solve_queue / closed_loop then take model_index.  None / empty removes the table."""
        hs = list(handles or [])
        arr = (c_void_p * max(len(hs), 1))(*[h._h for h in hs])
        check(self.lib.ampc_ilqr_plan_set_models(self._p, len(hs), arr))
        self._models = hs                   # (kept alive on this side as well)

    def _model_index(self, model_index, n):
        if model_index is None:
            return None
        return np.ascontiguousarray(np.broadcast_to(np.asarray(model_index, dtype=np.int32), (n,)))

    def closed_loop(self, init_obs, n_steps, cost_index=None, max_iter=50, surrogate=None, horizon=None,
                    model_index=None):
        """simulate() with IterativeLQR controllers for C episodes, device resident (ampc_ilqr_closed_loop):
        init_obs [C, nx]; returns dict(obs [C, n_steps+1, nx], ctrls [C, n_steps+1, nu], failed [C],
        steps [C], iterations [C]).  horizon [C] (optional): each episode's iLQR horizon, at most the
        plan's (ampc_ilqr_closed_loop_var)."""
        nx, nu = self.handle.nx, self.handle.nu
        init_obs = as_f64(init_obs).reshape(-1, nx)
        C = init_obs.shape[0]
        ci = None if cost_index is None else np.ascontiguousarray(np.broadcast_to(
            np.asarray(cost_index, dtype=np.int32), (C,)))
        hz = self._horizons(horizon, C)
        mi = self._model_index(model_index, C)
        out = {"obs": np.empty((C, n_steps + 1, nx)), "ctrls": np.empty((C, n_steps + 1, nu)),
               "failed": np.zeros(C, dtype=np.int32), "steps": np.zeros(C, dtype=np.int32),
               "iterations": np.zeros(C, dtype=np.int64)}
        check(self.lib.ampc_ilqr_closed_loop_var(
            self._p, surrogate._h if surrogate is not None else None, C, dptr(init_obs), iptr(ci), iptr(hz), iptr(mi),
            int(n_steps), int(max_iter), dptr(out["obs"]), dptr(out["ctrls"]), iptr(out["failed"]),
            iptr(out["steps"]), out["iterations"].ctypes.data_as(POINTER(ctypes.c_longlong))))
        return out

    def solve_queue(self, x0, uguess=None, cost_index=None, max_iter=50, gains=True, trajectories=True,
                    horizon=None, model_index=None):
        """P problems streamed through the plan's B slots (ampc_ilqr_solve_queue): a slot whose problem
        is finished takes the next one at the following iteration boundary, on the device.  x0 [P, nx];
        uguess [P, H, nu] or None (zeros); cost_index [P] or None (block 0).  Per-problem results are
        bit-identical to one-problem solves.  gains / trajectories = False skip those downloads.
        horizon [P] (optional): each problem's own horizon, at most the plan's (ampc_ilqr_solve_queue_var):
        arrays keep the plan's H as their stride, rows past a problem's horizon come back zero."""
        nx, nu, H = self.handle.nx, self.handle.nu, self.H
        x0 = as_f64(x0).reshape(-1, nx)
        P = x0.shape[0]
        ug = None if uguess is None else as_f64(uguess).reshape(P, H, nu)
        ci = None if cost_index is None else np.ascontiguousarray(np.broadcast_to(
            np.asarray(cost_index, dtype=np.int32), (P,)))
        out = {"converged": np.zeros(P, dtype=np.int32), "iters": np.zeros(P, dtype=np.int32),
               "status": np.zeros(P, dtype=np.int32), "objective": np.empty(P)}
        if trajectories:
            out["states"], out["ctrls"] = np.empty((P, H + 1, nx)), np.empty((P, H, nu))
        if gains:
            out["Ks"], out["ks"] = np.empty((P, H, nu, nx)), np.empty((P, H, nu))
        hz = self._horizons(horizon, P)
        mi = self._model_index(model_index, P)
        check(self.lib.ampc_ilqr_solve_queue_var(self._p, P, dptr(x0), dptr(ug), iptr(ci), iptr(hz), iptr(mi), int(max_iter),
                                                 dptr(out.get("states")), dptr(out.get("ctrls")), dptr(out.get("Ks")),
                                                 dptr(out.get("ks")), iptr(out["converged"]), iptr(out["iters"]),
                                                 iptr(out["status"]), dptr(out["objective"])))
        return out
