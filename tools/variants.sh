#!/bin/bash
# Build timing-experiment variants of the library (AMPC_X_* macros) into variants/ (git-ignored).
cd "$(dirname "$0")/.."
mkdir -p variants
build() { # name flags...
  local name=$1; shift
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-pass-failed -I include "$@" autompc_amd/csrc/autompc_hip.cpp -o variants/lib_$name.so &
}
build nomfma -DAMPC_X_NOMFMA
build noload -DAMPC_X_NOLOAD
build nocost -DAMPC_X_NOCOST
build nomfma_noload -DAMPC_X_NOMFMA -DAMPC_X_NOLOAD
build phasetime -DAMPC_X_PHASETIME
wait
ls -la variants
