#!/bin/bash
# VALU / SALU / LDS instruction counts of the rollout kernel per launch (c3 f64 headline bench).
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmcv; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_INT32 --output-format csv -d $O/a -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 2 $@ > $O/a.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmcv/*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "mppi_rollout_kernel" not in r["Kernel_Name"]: continue
        k = r["Counter_Name"]; agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    for k, (v, n) in sorted(agg.items()):
        print("   %-24s per launch %12.0f  per wave-step %7.1f" % (k, v / n, v / n / (2048 * 30)))
PY
