#!/bin/bash
# One GPU-box session for the record: parity tests, smoke, every bench line, rocprofv3 summaries.
# Usage (from repo root, via gpurun):  bash tools/gpu_round.sh r06
set -u
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -rs -p no:cacheprovider 2>&1 | tail -25 > $OUT/pytest_gpu.log      # (-rs: skip reasons in the log)
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1
b() { name=$1; shift; timeout 900 python bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
b c3_f64 
b c3_f64_driver --steps 20 --warmup 5 --no-cpu-baseline
b c3_f32 --precision f32 --no-cpu-baseline
b c2_f64 --workload c2 --cpu-seconds 10
b c3_f64_batch8 --batch 8 --steps 50 --warmup 5 --no-cpu-baseline
b arx_f64 --workload arx --cpu-seconds 10
b c1_sindy_f64 --workload c1 --cpu-seconds 10
b c4_ilqr_f64 --workload c4 --steps 5 --warmup 1
b c4_ilqr_f64_b256 --workload c4 --batch 256 --steps 5 --warmup 1 --no-cpu-baseline
b c4_ilqr_f64_b512 --workload c4 --batch 512 --steps 2 --warmup 1 --no-cpu-baseline
b c5_candidates_f64 --workload c5 --steps 2 --warmup 1
# launcher plumbing: `bench.py --gpus 2` starts its own two ranks; both mapped onto this box's one
# GPU (gloo for the barriers / all-gather; RCCL refuses two ranks on one device).  Not a performance number.
AMPC_BENCH_FORCE_DEVICE=0 AMPC_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_2rank_plumbing.json 2> $OUT/bench_2rank_plumbing.err
AMPC_BENCH_FORCE_DEVICE=0 AMPC_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --workload c5 --batch 8 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_2rank_c5_plumbing.json 2> $OUT/bench_2rank_c5_plumbing.err
timeout 300 python tools/dropin_rate.py > $OUT/dropin_rate.log 2>&1
timeout 300 python tools/dropin_breakdown.py > $OUT/dropin_breakdown.log 2>&1
timeout 600 python tools/model_axis_rate.py > $OUT/model_axis_rate.log 2>&1
timeout 900 python tools/world8_hosttime.py 8 64 200 > $OUT/world8_hosttime.log 2>&1
timeout 300 python tools/shape_groups_rate.py > $OUT/shape_groups_rate.log 2>&1
timeout 300 bash tools/fit_kernels.sh $TAG > /dev/null 2>&1
timeout 600 bash tools/c4_trace_iterations.sh 1024 1024 > $OUT/c4_iterations.log 2>&1
timeout 300 python tools/dropin_ilqr.py > $OUT/dropin_ilqr.log 2>&1
timeout 500 bash tools/dropin_kernels.sh $TAG > /dev/null 2>&1
timeout 600 python tools/jit_rate.py > $OUT/jit_rate.log 2>&1
timeout 600 python tools/ilqr_eval_rate.py 64 50 > $OUT/ilqr_eval_rate.log 2>&1
timeout 600 python tools/ilqr_eval_rate.py 256 30 >> $OUT/ilqr_eval_rate.log 2>&1
timeout 900 python tools/models_rate.py > $OUT/models_rate.log 2>&1
timeout 600 python tools/wide_ilqr_rate.py > $OUT/wide_ilqr_rate.log 2>&1
timeout 300 python tools/c4_queue_rate.py 1024 256 > $OUT/c4_queue_rate.log 2>&1
timeout 300 python tools/c4_queue_rate.py 8192 256 512 >> $OUT/c4_queue_rate.log 2>&1
timeout 300 python tools/c4_queue_groups.py 4096 512 1 2 >> $OUT/c4_queue_rate.log 2>&1
timeout 900 python tools/fuzz_gpu.py 400 31 > $OUT/fuzz_gpu.log 2>&1
timeout 900 python tools/fuzz_gpu.py 400 32 >> $OUT/fuzz_gpu.log 2>&1
timeout 120 python tools/validate_glibc_log.py > $OUT/glibc_log.log 2>&1
bash tools/gpu_profile.sh $TAG c3_f64_b1 --steps 600 --warmup 20 > $OUT/profile_c3.log 2>&1
bash tools/gpu_profile.sh $TAG c3_f32_b1 --precision f32 --steps 600 --warmup 20 > $OUT/profile_c3f32.log 2>&1
bash tools/gpu_profile.sh $TAG c2_f64_b1 --workload c2 --steps 2000 --warmup 50 > $OUT/profile_c2.log 2>&1
bash tools/gpu_profile.sh $TAG c4_f64_b256 --workload c4 --batch 256 --steps 2 --warmup 1 > $OUT/profile_c4.log 2>&1
bash tools/gpu_profile.sh $TAG c5_f64_b64 --workload c5 --steps 1 --warmup 1 > $OUT/profile_c5.log 2>&1
cat $OUT/pytest_gpu.log | tail -3; cat $OUT/smoke.log | tail -1
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"] or {"kernel_ms":0,"achieved":0,"frac":0}
        print("%-32s n_gpus=%d value=%9.1f ms/step=%8.3f kernel_ms=%.4f TF=%.1f frac=%.3f cpu=%s" % (f.split("/")[-1], d["n_gpus"], d["value"], d["ms_per_step"], r["kernel_ms"], r["achieved"], r["frac"], d.get("cpu_baseline",{}).get("value")))
    except Exception as e: print(f, "failed", e)
PY
