#!/bin/bash
# c4 (iLQR, 256 problems) with the product library and the variants named on the command line
cd "$GRAFT_REPO_ROOT"
run() {
  python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; k=r['per_iteration_kernel_ms']
print('%-14s solves/s=%8.1f  ms/step=%7.2f  riccati=%.3f iter=%.3f fwd=%.3f jac=%.3f  TF=%.2f' % ('$1', d['value'], d['ms_per_step'], k['riccati'], k['iter'], k['forward'], k['jacobian'], d['algorithmic_tflops']))"
}
run product
for v in "$@"; do AMPC_LIB=$GRAFT_REPO_ROOT/variants/lib_$v.so run $v; done
