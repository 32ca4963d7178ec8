#!/bin/bash
# Per-iteration kernel times of one iLQR queue run (rocprofv3 kernel trace of tools/c4_queue_rate.py P B):
# for every 4th iteration of the LAST queue run in the trace: wall span and the duration of each kernel.
#   bash tools/c4_trace_iterations.sh 1024 1024        (on the GPU box, from the repository root)
set -u
P=${1:-1024}; B=${2:-1024}
OUT=$GRAFT_REPO_ROOT/gpurun_out/c4_trace
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python $GRAFT_REPO_ROOT/tools/c4_queue_rate.py $P $B > $OUT/run.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
f = glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("<")[0].replace("void ampc::", "").replace("ampc::", "")
its, cur = [], None
for r in rows:
    n = short(r["Kernel_Name"])
    if n == "ilqr_queue_refill_kernel":
        cur = []
        its.append(cur)
    if cur is not None:
        cur.append((n, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
# the last run = the last block of consecutive iterations; a queue run ends where the gap to the next refill is large
runs, run = [], [its[0]]
for a, b in zip(its, its[1:]):
    if b[0][1] - a[-1][2] > 2_000_000:
        runs.append(run); run = []
    run.append(b)
runs.append(run)
run = runs[-1]
print("%d iterations in the last queue run, %.1f ms" % (len(run), (run[-1][-1][2] - run[0][0][1]) / 1e6))
for i in range(0, len(run), 4):
    it = run[i]
    span = (it[-1][2] - it[0][1]) / 1e3
    parts = {}
    for n, s, e in it:
        parts[n] = parts.get(n, 0) + (e - s) / 1e3
    print("it %3d  span %7.1f us  busy %7.1f  " % (i, span, sum(parts.values())) +
          "  ".join("%s %.0f" % (k.replace("_kernel", ""), v) for k, v in parts.items()))
PY
rm -rf $OUT/t
