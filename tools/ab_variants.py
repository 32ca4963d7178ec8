#!/usr/bin/env python3
"""Build timing-experiment variants of the library (AMPC_X_* macros, never defined in the product)
into variants/ and print the gpurun command that times them all with the headline bench.

    python tools/ab_variants.py name1:-DAMPC_X_FOO name2:-DAMPC_X_BAR,-DAMPC_X_BAZ ...
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autompc_amd.csrc.build import build   # noqa: E402

os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
for spec in sys.argv[1:]:
    name, _, flags = spec.partition(":")
    out = os.path.join(ROOT, "variants", "lib_%s.so" % name)
    build(force=True, verbose=False, extra_flags=[f for f in flags.split(",") if f], out=out)
    print("built", out, flush=True)
