#!/bin/bash
# rocprofv3 kernel trace + PMC passes (each in its own run: counters are never combined with
# other trace domains) of one bench workload.  Run on the GPU box from the repository root:
#   bash tools/gpu_profile.sh <tag> <key> <bench.py arguments...>
# e.g. bash tools/gpu_profile.sh r02 c3_f64_b1 --steps 100 --warmup 10
# Results: gpurun_out/<tag>/{rocprofv3_summary_<key>.txt, kernel_stats_<key>.csv, hbm_traffic.json}
set -u
TAG=$1; KEY=$2; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT/$KEY
export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --preheat 1.0 $@"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$KEY/trace -- $BENCH > $OUT/$KEY/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/$KEY/pmc_sq -- $BENCH > $OUT/$KEY/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$KEY/pmc_fetch -- $BENCH > $OUT/$KEY/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/$KEY/pmc_write -- $BENCH > $OUT/$KEY/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/$KEY/pmc_inst -- $BENCH > $OUT/$KEY/pmc_inst.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_profiles.py $OUT $KEY "python bench.py --no-cpu-baseline --no-extras --preheat 1.0 $*" | head -60
# keep only the summaries in what travels back (the raw CSVs are large)
rm -rf $OUT/$KEY
