"""The restatement of glibc's log() (oracle/glibc_log.c, csrc/glibc_log.hpp) against the C library
it claims to reproduce, on the machine it runs on:
  * the build log() resolves to in this process (what numpy's legacy_gauss calls): 3.5 * 10^7 arguments
  * both builds glibc carries behind its IFUNC, entered directly through their addresses in the
    loaded libm (found by disassembly for Ubuntu GLIBC 2.35-0ubuntu3.11; skipped for another libm)
    python tools/validate_glibc_log.py"""
import ctypes
import os
import platform
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import glibc_log                      # noqa: E402

t = glibc_log.locate()
if t is None:
    print("no glibc __log_data in the loaded libm")
    sys.exit(0)
variant = glibc_log.probe(t, 2_000_000)
print("libc:", platform.libc_ver(), " log() is build", variant, "(1 = FMA, 2 = plain)")
rng = np.random.default_rng(0)
a, b = 2 * rng.random(2 * 10 ** 7) - 1, 2 * rng.random(2 * 10 ** 7) - 1
r2 = a * a + b * b
sets = {"polar r2": r2[(r2 < 1) & (r2 > 0)], "near 1": 0.93 + 0.14 * rng.random(10 ** 7),
        "full range": np.exp(rng.uniform(-700, 700, 10 ** 7))}
for name, x in sets.items():
    for v in (1, 2):
        _, bad = glibc_log.restated_log(x, t, v)
        print("%-11s %9d arguments  build %d restated: %d mismatches vs log()" % (name, x.size, v, bad))
L = glibc_log.lib()
L.glibc_log_restated.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.c_int]
L.glibc_log_restated.restype = ctypes.c_double
if "2.35" in platform.libc_ver()[1]:
    base = None
    for line in open("/proc/self/maps"):
        if "libm.so.6" in line:
            base = int(line.split("-")[0], 16)
            break
    FT = ctypes.CFUNCTYPE(ctypes.c_double, ctypes.c_double)
    xs = np.concatenate([rng.random(200000), 0.93 + 0.14 * rng.random(200000), np.exp(rng.uniform(-700, 700, 50000))])
    tp = t.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    for off, name, want in ((0x29200, "__log_sse2", 2), (0x7fb30, "__log_avx", 2), (0x76660, "__log_fma", 1)):
        f = FT(base + off)
        bad = {1: 0, 2: 0}
        for x in xs:
            r = f(x)
            for v in (1, 2):
                if L.glibc_log_restated(float(x), tp, v) != r:
                    bad[v] += 1
        print("%-11s (libm + 0x%x) %d arguments: mismatches build-1 restatement %d, build-2 restatement %d  -> %s"
              % (name, off, len(xs), bad[1], bad[2], "OK" if bad[want] == 0 else "MISMATCH"))
