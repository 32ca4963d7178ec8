#!/bin/bash
# Per-kernel breakdown of the drop-in MPPI.run() call in its two noise modes (rocprofv3 kernel trace of
# tools/dropin_rate.py): which launches the parity-graded numpy-stream mode adds to a call.
# Run on the GPU box from the repository root:  bash tools/dropin_kernels.sh r05
set -u
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT/dropin_trace
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dropin_trace -- python $GRAFT_REPO_ROOT/tools/dropin_rate.py > $OUT/dropin_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY > $OUT/dropin_kernels.txt
import csv, glob
f = glob.glob("$OUT/dropin_trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("# rocprofv3 --kernel-trace --stats of python tools/dropin_rate.py (c3, c2, arx MPPI.run in the device / numpy / numpy_host")
print("# noise modes, then one-problem IterativeLQR.run): calls, average ns, share of the GPU time")
for r in rows:
    print("%-110s calls=%-7s avg_ns=%-9.0f pct=%s" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
tail -12 $OUT/dropin_trace.log >> $OUT/dropin_kernels.txt
rm -rf $OUT/dropin_trace $OUT/dropin_trace.log
head -30 $OUT/dropin_kernels.txt
