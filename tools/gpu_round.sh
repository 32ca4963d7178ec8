#!/bin/bash
# One GPU-box session for the record: parity tests, smoke, bench, rocprofv3 summaries.
# Usage (from repo root, via gpurun):  bash tools/gpu_round.sh r01
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1
timeout 900 python bench.py > $OUT/bench_c3_f64.json 2> $OUT/bench_c3_f64.err
timeout 300 python bench.py --precision f32 --no-cpu-baseline > $OUT/bench_c3_f32.json 2> $OUT/bench_c3_f32.err
timeout 300 python bench.py --workload c2 > $OUT/bench_c2_f64.json 2> $OUT/bench_c2_f64.err
timeout 300 python bench.py --batch 8 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_c3_f64_batch8.json 2> $OUT/bench_c3_f64_batch8.err
timeout 300 python bench.py --workload arx > $OUT/bench_arx_f64.json 2> $OUT/bench_arx_f64.err
timeout 300 python bench.py --workload c1 > $OUT/bench_c1_sindy_f64.json 2> $OUT/bench_c1_sindy_f64.err
timeout 600 python bench.py --workload c4 --steps 3 --warmup 1 > $OUT/bench_c4_ilqr_f64.json 2> $OUT/bench_c4.err
timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_c5_candidates_f64.json 2> $OUT/bench_c5.err
# launcher plumbing: the driver's N>1 command line with two ranks mapped onto this box's one GPU
# (gloo for the barriers; RCCL refuses two ranks on one device).  Not a performance number.
AMPC_BENCH_FORCE_DEVICE=0 AMPC_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 50 --warmup 5 > $OUT/bench_2rank_plumbing.json 2> $OUT/bench_2rank_plumbing.err
AMPC_BENCH_FORCE_DEVICE=0 AMPC_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --workload c5 --batch 8 --steps 1 --warmup 1 > $OUT/bench_2rank_c5_plumbing.json 2> $OUT/bench_2rank_c5_plumbing.err
BENCH="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline"   # default 200 timed + 20 warm-up solves, as the headline run
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -- $BENCH > $OUT/prof_trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_pmc_sq -- $BENCH > $OUT/prof_pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_pmc_fetch -- $BENCH > $OUT/prof_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_pmc_write -- $BENCH > $OUT/prof_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS --output-format csv -d $OUT/prof_pmc_lds -- $BENCH > $OUT/prof_pmc_lds.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT -name "*.csv" | head -30
cat $OUT/pytest_gpu.log | tail -3; cat $OUT/smoke.log | tail -1
python tools/summarize_profiles.py $OUT > $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_summary.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"] or {"kernel_ms":0,"achieved":0,"frac":0}
        print("%-28s value=%8.1f ms/step=%.3f kernel_ms=%.3f TF=%.1f frac=%.3f cpu=%s" % (f.split("/")[-1], d["value"], d["ms_per_step"], r["kernel_ms"], r["achieved"], r["frac"], d.get("cpu_baseline",{}).get("value")))
    except Exception as e: print(f, "failed", e)
PY
