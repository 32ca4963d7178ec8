#!/bin/bash
# rocprofv3 kernel statistics of the lockstep MLP fit (tools/model_axis_rate.py with 6 epochs): which kernels a step
# is made of, how long each runs.  bash tools/fit_kernels.sh <tag>      (on the GPU box, from the repository root)
set -u
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fit_trace -- python $GRAFT_REPO_ROOT/tools/model_axis_rate.py 8 6 > $OUT/fit_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
f = glob.glob("$OUT/fit_trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
out = ["# rocprofv3 --kernel-trace --stats of python tools/model_axis_rate.py 8 6: eight 2x256 models, 6 epochs x 125 steps in lockstep",
       "# (750 lockstep steps + 2 capture warm-ups; the ~254-call rows belong to the 2-epoch sequential baseline fit;",
       "#  ampc:: rows to data generation and the closed-loop evaluation).  calls, average ns, share of the GPU time"]
for r in rows[:45]:
    out.append("%-110s calls=%-6s avg_ns=%-9s pct=%s" % (r["Name"][:110], r["Calls"], r["AverageNs"].split(".")[0], r["Percentage"][:6]))
open("$OUT/fit_kernels.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:40]))
PY
grep -E "fit_s|us_per|sequential" $OUT/fit_trace.log
rm -rf $OUT/fit_trace
