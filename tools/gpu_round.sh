#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof summary.  Run via gpurun from repo root.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
timeout 600 python bench.py --steps 200 --warmup 20 > gpurun_out/bench_f64.json 2> gpurun_out/bench_f64.err
timeout 300 python bench.py --steps 200 --warmup 20 --precision f32 --no-cpu-baseline > gpurun_out/bench_f32.json 2> gpurun_out/bench_f32.err
timeout 300 python bench.py --steps 200 --warmup 20 --workload c2 --no-cpu-baseline > gpurun_out/bench_c2_f64.json 2> gpurun_out/bench_c2.err
for mt in 1 2; do
  AMPC_MT=$mt timeout 300 python bench.py --steps 100 --warmup 10 --batch 8 --no-cpu-baseline > gpurun_out/bench_b8_mt$mt.json 2>> gpurun_out/bench_b8.err
done
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_c3 -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_c3.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_c3 -name "*kernel_stats*" | head -3
cat gpurun_out/pytest_gpu.log gpurun_out/smoke.log | tail -20
head -c 2500 gpurun_out/bench_f64.json; echo; tail -3 gpurun_out/bench_f64.err
