#!/bin/bash
# iLQR parity tests + the drop-in timing of the H = 50 golden problems + the c4 bench line
timeout 900 python -m pytest tests/test_gpu_ilqr.py tests/test_gpu_properties.py tests/test_linear_models.py tests/test_gpu_sindy.py -m gpu -x -q 2>&1 | tail -8
timeout 300 python tools/dropin_ilqr.py 2>&1 | tail -8
timeout 600 python bench.py --workload c4 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('c4', round(d['value'],1), 'solves/s', 'ms/step', round(d['ms_per_step'],3), 'alg TF', round(d['algorithmic_tflops'],2), r['per_iteration_kernel_ms'])"
