"""ctypes binding of oracle/glibc_log.c (the CPU restatement of glibc's log(); oracle, test-only).

Why log() is on the path: the reference's MPPI noise is np.random.normal from numpy's global
legacy generator (autompc/control/mppi.py:16-24, :126), whose polar method evaluates
sqrt(-2 log(r2) / r2) with the C library's log() -- see the header of glibc_log.c.
"""
import ctypes

import numpy as np

from .build import build

TABLE_DOUBLES = 2 + 5 + 11 + 256 + 256
_dp = ctypes.POINTER(ctypes.c_double)
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.glibc_log_locate.argtypes = [_dp]
        L.glibc_log_probe.argtypes = [_dp, ctypes.c_long]
        L.glibc_log_compare.argtypes = [_dp, ctypes.c_long, _dp, ctypes.c_int, _dp]
        L.glibc_log_compare.restype = ctypes.c_long
        _lib = L
    return _lib


def locate():
    """The host libm's __log_data (530 doubles), or None when it is not glibc's table."""
    t = np.zeros(TABLE_DOUBLES)
    return t if lib().glibc_log_locate(t.ctypes.data_as(_dp)) == 0 else None


def probe(table, n=200000):
    """1: the host's log() is glibc's FMA build, 2: the plain build, 0: neither."""
    return int(lib().glibc_log_probe(table.ctypes.data_as(_dp), n))


def restated_log(x, table, variant):
    """(values, number of bitwise mismatches against the host's log())."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    bad = lib().glibc_log_compare(x.ctypes.data_as(_dp), x.size, table.ctypes.data_as(_dp), int(variant),
                                  out.ctypes.data_as(_dp))
    return out, int(bad)
