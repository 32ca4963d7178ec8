#!/bin/bash
# Quick perf loop on the GPU box: parity tests (fast) + bench variants.  Run via gpurun.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
run() { # name, args...
  local name=$1; shift
  timeout 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/$name.json 2> gpurun_out/$name.err || tail -3 gpurun_out/$name.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$name.json")); r=d["roofline"]
    print("%-14s value=%8.1f ms/step=%.3f kernel_ms=%.3f upd=%.3f TF=%.1f frac=%.3f wgs=%d spw=%d" % ("$name", d["value"], d["ms_per_step"], r["kernel_ms"], r["update_kernel_ms"], r["achieved"], r["frac"], r["workgroups"], r["samples_per_workgroup"]))
except Exception as e:
    print("$name failed", e)
PY
}
run c3_f64 --steps 200 --warmup 20
run c3_f32 --steps 200 --warmup 20 --precision f32
run c2_f64 --steps 200 --warmup 20 --workload c2
run arx_f64 --steps 200 --warmup 20 --workload arx
AMPC_MT=1 run c3_f64_b8_mt1 --steps 50 --warmup 5 --batch 8
AMPC_MT=2 run c3_f64_b8_mt2 --steps 50 --warmup 5 --batch 8
AMPC_MT=1 run c3_f32_b8_mt1 --steps 50 --warmup 5 --batch 8 --precision f32
AMPC_MT=2 run c3_f32_b8_mt2 --steps 50 --warmup 5 --batch 8 --precision f32
AMPC_MT=4 run c3_f32_b8_mt4 --steps 50 --warmup 5 --batch 8 --precision f32
