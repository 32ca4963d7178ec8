"""Build libautompc_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OUT = os.path.join(PKG, "libautompc_hip.so")
SOURCES = ["autompc_hip.cpp"]
HEADERS = ["mlp_tile.hpp", "mlp_kernels.hpp", "mppi_kernels.hpp", "rng_kernels.hpp",
           os.path.join(ROOT, "include", "autompc_hip.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(HERE, s) for s in SOURCES] + \
           [h if os.path.isabs(h) else os.path.join(HERE, h) for h in HEADERS] + [__file__]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
    if not force and not _stale():
        return OUT
    cmd = [_hipcc(), "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-pass-failed", "-I", os.path.join(ROOT, "include")]
    cmd += [os.path.join(HERE, s) for s in SOURCES] + ["-o", OUT + ".tmp"]
    if verbose:
        print("[autompc_amd] " + " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
