"""The C library's log() -- what numpy's legacy normal generator (the reference's MPPI noise,
mppi.py:16-24, :126) calls -- against its restatements: oracle/glibc_log.c (CPU) and the library's
own host-side proof (ampc_legacy_log_mode), which gates the device path (csrc/glibc_log.hpp)."""
import math

import numpy as np
import pytest

from oracle import glibc_log


def _polar_r2(n, seed):
    """Arguments exactly as legacy_gauss forms them: r2 = x1^2 + x2^2 with x = 2 u - 1, u a
    53-bit uniform, accepted when 0 < r2 < 1."""
    rng = np.random.RandomState(seed)
    x1 = 2.0 * rng.random_sample(n) - 1.0
    x2 = 2.0 * rng.random_sample(n) - 1.0
    r2 = x1 * x1 + x2 * x2
    return r2[(r2 < 1.0) & (r2 != 0.0)]


def test_table_is_found_and_one_build_reproduces_log():
    t = glibc_log.locate()
    if t is None:
        pytest.skip("the host C library is not glibc >= 2.28 (no __log_data)")
    assert t[0] == float.fromhex("0x1.62e42fefa3800p-1") and t[7] == -0.5
    variant = glibc_log.probe(t)
    assert variant in (1, 2)
    for x in (_polar_r2(6_000_000, 1), 0.93 + 0.14 * np.random.default_rng(2).random(2_000_000),
              np.exp(np.random.default_rng(3).uniform(-700, 700, 1_000_000)),
              np.array([1.0, 0.5, 2.0, 1.0 - 2.0 ** -53, 1.0 + 2.0 ** -52, 2.0 ** -1022, 5e-324,
                        1.7976931348623157e308, 0.9375, 1.0647])):
        y, bad = glibc_log.restated_log(x, t, variant)       # compared in C with log() of libm
        assert bad == 0
        # (np.log on arrays is numpy's own SIMD routine, not the C library's; math.log is libm's)
        np.testing.assert_array_equal(y[:20000], [math.log(v) for v in x[:20000]])
    # the two builds really differ: the other one must disagree somewhere on the same arguments
    _, bad_other = glibc_log.restated_log(_polar_r2(2_000_000, 4), t, 3 - variant)
    assert bad_other > 0


def test_library_proof_agrees_with_the_oracle():
    """The HIP library runs the same proof on the host at first use; its verdict decides whether
    MPPI's default noise mode may draw numpy's stream on the device."""
    from autompc_amd import _lib
    t = glibc_log.locate()
    mode = _lib.legacy_log_mode()
    if t is None:
        assert mode == 0
    else:
        assert mode == glibc_log.probe(t)
