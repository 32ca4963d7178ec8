"""Timeline of one steady-state control step of the drop-in MPPI.run() from a rocprofv3 kernel
trace (kernel_trace.csv): every kernel of the call with its queue, start and end relative to the
call's first kernel.  Usage: python tools/dropin_timeline.py <kernel_trace.csv> [call index]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 200
# a call = the kernels between two consecutive rollouts
roll = [i for i, r in enumerate(rows) if "mppi_rollout" in r["Kernel_Name"]]
lo, hi = roll[which - 1] + 1, roll[which + 1]
t0 = int(rows[roll[which - 1]]["Start_Timestamp"])
for r in rows[roll[which - 1]:hi + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("q%-3s %9.1f %9.1f  %7.1f us  %s" % (r["Queue_Id"], s / 1e3, e / 1e3, (e - s) / 1e3, r["Kernel_Name"][:70]))
