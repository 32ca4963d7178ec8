#!/bin/bash
# The headline subset of gpu_round.sh (tests, smoke, c3 / c1 / c4 bench lines, drop-in rates, c3 + c4 profiles):
# bash tools/gpu_round_short.sh   (results into gpurun_out/r04, then python tools/collect_profiles.py r04)
set -u
TAG=r04
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1
b() { name=$1; shift; timeout 900 python bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
b c3_f64
b c3_f64_driver --steps 20 --warmup 5 --no-cpu-baseline
b c3_f32 --precision f32 --no-cpu-baseline
b c3_f64_batch8 --batch 8 --steps 50 --warmup 5 --no-cpu-baseline
b c1_sindy_f64 --workload c1 --cpu-seconds 10
b c4_ilqr_f64 --workload c4 --steps 3 --warmup 1
timeout 300 python tools/dropin_rate.py > $OUT/dropin_rate.log 2>&1
timeout 300 python tools/dropin_ilqr.py > $OUT/dropin_ilqr.log 2>&1
bash tools/gpu_profile.sh $TAG c3_f64_b1 --steps 600 --warmup 20 > $OUT/profile_c3.log 2>&1
bash tools/gpu_profile.sh $TAG c3_f32_b1 --precision f32 --steps 600 --warmup 20 > $OUT/profile_c3f32.log 2>&1
bash tools/gpu_profile.sh $TAG c4_f64_b256 --workload c4 --steps 2 --warmup 1 > $OUT/profile_c4.log 2>&1
tail -3 $OUT/pytest_gpu.log; tail -1 $OUT/smoke.log
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"] or {"kernel_ms":0,"achieved":0,"frac":0}
        print("%-32s value=%9.1f ms/step=%8.3f kernel_ms=%.4f frac=%.3f cpu=%s" % (f.split("/")[-1], d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], d.get("cpu_baseline",{}).get("value")))
    except Exception as e: print(f, "failed", e)
PY
