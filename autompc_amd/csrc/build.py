"""Build libautompc_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

Seven translation units compiled in parallel and linked into one shared library: api.cpp (the C
ABI and host logic) plus launch_{mlp,mppi,ilqr}.cpp once per precision (-DAMPC_T=double|float).
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
HEADERS = ["host_common.hpp", "shapes.hpp", "legacy_rng_kernels.hpp", "mlp_tile.hpp", "mlp_kernels.hpp", "mppi_kernels.hpp", "ilqr_kernels.hpp", "ilqr_ls4.hpp",
           "rng_kernels.hpp", "sindy_kernels.hpp", "score_kernels.hpp", os.path.join(ROOT, "include", "autompc_hip.h")]
# (object name, source file, extra flags)
UNITS = [("api", "api.cpp", [])] + [
    ("%s_%s" % (fam, t), "launch_%s.cpp" % fam, ["-DAMPC_T=%s" % t] + (["-DAMPC_T_IS_F64=1"] if t == "double" else []))
    for fam in ("mlp", "mppi", "ilqr") for t in ("double", "float")]
SOURCES = sorted({u[1] for u in UNITS})


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return "hipcc"


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(HERE, f) for f in SOURCES] + [__file__] + [h if os.path.isabs(h) else os.path.join(HERE, h) for h in HEADERS]
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
        name, source, flags = unit
        obj = os.path.join(OBJ, "%s.%s.o" % (tag, name))
        cmd = base + flags + ["-c", os.path.join(HERE, source), "-o", obj]
        if verbose:
            print("[autompc_amd] hipcc %s %s -> %s" % (source, " ".join(flags), os.path.basename(obj)), flush=True)
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
