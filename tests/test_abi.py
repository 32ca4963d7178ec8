"""The C-ABI library loads on a CPU-only host and exports every symbol include/autompc_hip.h
declares (no compute calls here: there is no GPU in the build container)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "autompc_hip.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ampc_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib_path():
    from autompc_amd.csrc.build import build
    return build(verbose=False)          # no-op when the library is up to date


def test_every_declared_symbol_is_exported_and_bound(lib_path):
    from autompc_amd import _lib
    names = _declared()
    assert len(names) >= 20
    lib = ctypes.CDLL(lib_path)
    for n in names:
        assert hasattr(lib, n), "libautompc_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "autompc_amd/_lib.py has no prototype for %s" % n
    assert sorted(_lib.SIGNATURES) == names, "prototype table and header disagree"


def test_library_loads_and_reports_no_device_here(lib_path):
    from autompc_amd import _lib
    lib = _lib.load()
    assert lib.ampc_version() >= 100
    if lib.ampc_device_count() == 0:                    # the CPU container
        with pytest.raises(_lib.AmpcError):
            _lib.Handle(0, "f64")                       # fails loudly: no CPU fallback


def test_product_path_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under autompc_amd/ may import it."""
    pkg = os.path.join(ROOT, "autompc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), \
                    "%s imports the oracle" % os.path.join(dirpath, f)


def test_host_types_and_costs_known_answers():
    """The reference's own pinned cost values (tests/test_costs.py:192-205): SumCost of three
    QuadCosts at obs [-1, 1] -> 8, gradient [-4, 12], Hessian diag(4, 12)."""
    from autompc_amd import QuadCost, System
    system = System(["x", "y"], ["u"])
    c1 = QuadCost(system, np.eye(2), np.eye(1), np.eye(2), np.zeros(2))
    c2 = QuadCost(system, np.diag([1.0, 2.0]), 0.1 * np.eye(1), np.diag([1.0, 3.0]), np.zeros(2))
    c3 = QuadCost(system, np.diag([0.0, 3.0]), 0.5 * np.eye(1), np.diag([3.0, 0.0]), np.array([1.0, 0.0]))
    total = c1 + c2 + c3
    assert [type(c) for c in total.costs] == [QuadCost] * 3 and total.costs[2] is c3
    obs = np.array([-1, 1])
    assert total.eval_obs_cost(obs) == 8
    res, jac, hess = total.eval_obs_cost_hess(obs)
    assert res == 8 and (jac == np.array([-4, 12])).all() and (hess == np.diag([4, 12])).all()
    assert (c1 + c2).is_quad and (c1 + c2).has_goal and not (c1 + c3).is_quad
    assert not (c1 + c3).has_goal and (c1 + c3).is_convex and (c1 + c3).is_twice_diff


def test_header_is_plain_c99_and_links_from_c(lib_path, tmp_path):
    """The boundary is a C ABI: the header compiles as strict C99 and a C program (no C++, no
    torch types) links against the library and calls the device-free entry points."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "abi_probe.c"
    src.write_text(
        '#include <stdio.h>\n#include "autompc_hip.h"\n'
        "int main(void) {\n"
        "  ampc_handle* h = 0;\n"
        "  int v = ampc_version();\n"
        "  int n = ampc_device_count();\n"
        "  int rc = n > 0 ? 0 : ampc_create(0, 0, 0, &h);\n"          # must fail without a GPU
        '  printf("%d %d %d %s\\n", v, n, rc, ampc_last_error());\n'
        "  return v >= 100 ? 0 : 1;\n}\n")
    exe = tmp_path / "abi_probe"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I",
                    os.path.join(ROOT, "include"), str(src), "-o", str(exe), lib_path,
                    "-Wl,-rpath," + os.path.dirname(lib_path), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split(None, 3)
    assert int(out[0]) >= 100
    if int(out[1]) == 0:
        assert int(out[2]) != 0 and len(out) == 4 and out[3].strip()     # loud failure + message
