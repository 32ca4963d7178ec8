#!/bin/bash
# Time the product library and every variants/lib_*.so named on the command line with the headline
# bench (c3 f64, 200 steps) -- run on the GPU box:  bash tools/ab_run.sh [extra bench args --] name1 name2 ...
cd "$GRAFT_REPO_ROOT"
ARGS="--no-cpu-baseline --no-extras --steps 200 --warmup 20"
run() {
  python bench.py $ARGS $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-22s value=%8.1f  ms/step=%.4f  kernel_ms=%.4f  frac=%.4f' % ('$1', d['value'], d['ms_per_step'], r['kernel_ms'], r['frac']))"
}
EXTRA=""
if [ "$1" == "--extra" ]; then EXTRA="$2"; shift; shift; fi
run product
for v in "$@"; do AMPC_LIB=$GRAFT_REPO_ROOT/variants/lib_$v.so run $v; done
run product
