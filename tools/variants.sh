#!/bin/bash
# Build timing-experiment variants of the library into variants/ (git-ignored; they travel to the GPU
# box with gpurun).  AMPC_X_* macros are experiments only and are never defined in the product build.
#   lib_phasetime.so  per-phase s_memtime marks in the rollout kernel      (tools/phasetime.py)
#   lib_nomfma.so     MFMAs replaced by one scalar FMA: everything-but-MFMA time
#   mfma_peak         f64 / f32 MFMA issue ceiling microbenchmark
cd "$(dirname "$0")/.."
mkdir -p variants
python - <<'PY'
from autompc_amd.csrc.build import build
import os
root = os.getcwd()
build(force=True, verbose=False, extra_flags=["-DAMPC_X_PHASETIME"], out=os.path.join(root, "variants", "lib_phasetime.so"))
build(force=True, verbose=False, extra_flags=["-DAMPC_X_NOMFMA"], out=os.path.join(root, "variants", "lib_nomfma.so"))
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.cpp -o variants/mfma_peak 2>/dev/null
ls -la variants
