"""Build libautompc_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

autompc_hip.cpp is compiled as seven translation units in parallel -- the C API plus one unit
per (heavy kernel family, precision) -- and linked into one shared library.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OUT = os.path.join(PKG, "libautompc_hip.so")
OBJ = os.path.join(HERE, "build")
SOURCE = os.path.join(HERE, "autompc_hip.cpp")
HEADERS = ["mlp_tile.hpp", "mlp_kernels.hpp", "mppi_kernels.hpp", "ilqr_kernels.hpp",
           "rng_kernels.hpp", "sindy_kernels.hpp", "score_kernels.hpp", os.path.join(ROOT, "include", "autompc_hip.h")]
UNITS = [("main", ["-DAMPC_TU_MAIN"])] + [
    ("f%d_%s" % (fam, t), ["-DAMPC_TU_FAMILY=%d" % fam, "-DAMPC_TU_T=%s" % t] +
     (["-DAMPC_TU_F64=1"] if t == "double" else []))
    for fam in (1, 2, 3) for t in ("double", "float")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return "hipcc"


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [SOURCE, __file__] + [h if os.path.isabs(h) else os.path.join(HERE, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True, extra_flags=(), out=None):
    out = out or OUT
    if not force and not extra_flags and out == OUT and not _stale():
        return out
    os.makedirs(OBJ, exist_ok=True)
    base = [_hipcc(), "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
            "-Wno-pass-failed", "-I", os.path.join(ROOT, "include")] + list(extra_flags)
    tag = os.path.basename(out)

    def compile_unit(unit):
        name, flags = unit
        obj = os.path.join(OBJ, "%s.%s.o" % (tag, name))
        cmd = base + flags + ["-c", SOURCE, "-o", obj]
        if verbose:
            print("[autompc_amd] hipcc %s -> %s" % (" ".join(flags), os.path.basename(obj)), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    workers = min(len(UNITS), max(1, os.cpu_count() or 1))
    with concurrent.futures.ThreadPoolExecutor(workers) as pool:
        objs = list(pool.map(compile_unit, UNITS))
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out + ".tmp"]
    subprocess.run(link, check=True)
    os.replace(out + ".tmp", out)
    if verbose:
        print("[autompc_amd] linked %s" % out, flush=True)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
