"""The shape plugins compiled at run time (csrc/jit_host.hpp) are cached under a hash of their
sources: the hashed list must cover every file the plugin's translation units include, or a stale
plugin could be loaded after a header changed."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "autompc_amd", "csrc")


def _includes(path, seen):
    for name in re.findall(r'^\s*#include\s+"([^"]+)"', open(path).read(), flags=re.M):
        full = os.path.normpath(os.path.join(os.path.dirname(path), name))
        if not os.path.exists(full):                       # "autompc_hip.h" comes from -I include/
            full = os.path.join(ROOT, "include", name)
        if full not in seen and os.path.exists(full):
            seen.add(full)
            _includes(full, seen)
    return seen


def test_hashed_source_list_covers_the_plugin_translation_units():
    units = ["launch_mppi.cpp", "launch_mlp.cpp", "launch_ilqr.cpp", "jit_plugin.cpp"]
    needed = set(os.path.join(CSRC, u) for u in units)
    for u in units:
        _includes(os.path.join(CSRC, u), needed)
    text = open(os.path.join(CSRC, "jit_host.hpp")).read()
    block = text[text.index("source_files()"):]
    block = block[:block.index("};")]
    listed = {os.path.normpath(os.path.join(CSRC, f)) for f in re.findall(r'"([^"]+)"', block)}
    missing = sorted(os.path.relpath(p, ROOT) for p in needed - listed)
    assert not missing, "jit_host.hpp source_files() lacks: %s" % missing
    assert all(os.path.exists(p) for p in listed)


def test_product_headers_read_no_experiment_flags():
    """Timing-experiment build flags (-DAMPC_X_*) are read in csrc/probe.hpp only."""
    for name in os.listdir(CSRC):
        if name.endswith((".hpp", ".cpp")) and name != "probe.hpp":
            assert "AMPC_X_" not in open(os.path.join(CSRC, name)).read(), name


def test_ilqr_plugin_unit_compiles_for_a_deep_unregistered_shape(tmp_path):
    """The iLQR translation unit of a shape plugin, cross-compiled here the way csrc/jit_host.hpp does on
    the user's box, for a shape whose hidden -> hidden layers run in a RUN-TIME layer loop (three hidden
    layers, 192 wide): code generation problems of the line-search kernels that only show for such shapes
    (a wave-uniform value the compiler can no longer prove uniform) must not wait for a GPU box to show."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc here")
    cmd = [hipcc, "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed",
           "-I", os.path.join(ROOT, "include"), "-DAMPC_JIT_PLUGIN", "-DAMPC_T=double", "-DAMPC_T_IS_F64=1",
           "-DAMPC_JIT_NX=12", "-DAMPC_JIT_NU=3", "-DAMPC_JIT_NO=12", "-DAMPC_JIT_NH=3", "-DAMPC_JIT_HPAD=192",
           "-c", os.path.join(CSRC, "launch_ilqr.cpp"), "-o", str(tmp_path / "ilqr_plugin.o")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
