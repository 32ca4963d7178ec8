"""Build libautompc_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

Ten translation units compiled in parallel and linked into one shared library: api.cpp + api_{model,mppi,ilqr}.cpp (the C ABI
and host logic by family) plus launch_{mlp,mppi,ilqr}.cpp once per precision (-DAMPC_T=double|float).
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
import glob                                                   # noqa: E402


def _headers():
    """Every header a translation unit may include: all of csrc/*.hpp and the public C header."""
    return sorted(glob.glob(os.path.join(HERE, "*.hpp"))) + [os.path.join(ROOT, "include", "autompc_hip.h")]
# (object name, source file, extra flags)
UNITS = [("api", "api.cpp", []), ("api_model", "api_model.cpp", []), ("api_mppi", "api_mppi.cpp", []),
         ("api_ilqr", "api_ilqr.cpp", [])] + [
    ("%s_%s" % (fam, t), "launch_%s.cpp" % fam, ["-DAMPC_T=%s" % t] + (["-DAMPC_T_IS_F64=1"] if t == "double" else []))
    for fam in ("mlp", "mppi", "ilqr") for t in ("double", "float")]
SOURCES = sorted({u[1] for u in UNITS})


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return "hipcc"


def _obj_path(out, name):
    return os.path.join(OBJ, "%s.%s.o" % (os.path.basename(out), name))


def _unit_stale(out, unit):
    """An object is stale when it is older than its source, any header or this script.  Judged per
    OBJECT, not per library: a hand-made relink must never hide objects that were compiled against
    older headers (struct layouts are shared between the translation units)."""
    obj = _obj_path(out, unit[0])
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [os.path.join(HERE, unit[1]), __file__] + _headers()
    return any(os.path.getmtime(d) > t for d in deps)


def _stale(out=OUT):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(_unit_stale(out, u) or os.path.getmtime(_obj_path(out, u[0])) > t for u in UNITS)


def build(force=False, verbose=True, extra_flags=(), out=None):
    out = out or OUT
    if not force and not extra_flags and out == OUT and not _stale():
        return out
    os.makedirs(OBJ, exist_ok=True)
    todo = [u for u in UNITS if force or extra_flags or _unit_stale(out, u)]
    base = [_hipcc(), "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
            "-Wno-pass-failed", "-I", os.path.join(ROOT, "include")] + list(extra_flags)
    tag = os.path.basename(out)

    def compile_unit(unit):
        name, source, flags = unit
        obj = _obj_path(out, name)
        cmd = base + flags + ["-c", os.path.join(HERE, source), "-o", obj]
        if verbose:
            print("[autompc_amd] hipcc %s %s -> %s" % (source, " ".join(flags), os.path.basename(obj)), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    workers = min(len(UNITS), max(1, os.cpu_count() or 1))
    with concurrent.futures.ThreadPoolExecutor(workers) as pool:
        list(pool.map(compile_unit, todo))
    objs = [_obj_path(out, u[0]) for u in UNITS]
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out + ".tmp"]
    subprocess.run(link, check=True)
    os.replace(out + ".tmp", out)
    if verbose:
        print("[autompc_amd] linked %s" % out, flush=True)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
