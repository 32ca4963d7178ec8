"""Condense the rocprofv3 CSVs of one tools/gpu_profile.sh run (one workload) into a text summary
(per-kernel averages of the kernel trace and of every collected counter, derived MFMA-busy fraction
and HBM bytes) and merge its HBM traffic into <out>/hbm_traffic.json under <key>.

    python tools/summarize_profiles.py <out_dir> <key> "<command line that was profiled>"
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out, key, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
base = os.path.join(out, key)
lines = ["# rocprofv3 summary of `%s` -- averages per launch" % cmd]
for f in glob.glob(base + "/trace/*/*_kernel_stats.csv"):
    shutil.copy(f, os.path.join(out, "kernel_stats_%s.csv" % key))
    lines.append("\n## kernel-trace --stats")
    for row in csv.DictReader(open(f)):
        lines.append("%-96s calls=%-5s avg_ns=%-10.0f pct=%s" % (row["Name"][:96], row["Calls"],
                                                               float(row["AverageNs"]), row["Percentage"]))
vals = collections.defaultdict(dict)
for d in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_inst"):
    for f in glob.glob(base + "/" + d + "/*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            for c, x in v.items():
                vals[k][c] = sum(x) / len(x)
traffic = {}
lines.append("\n## PMC (separate passes), average per launch")
for k, v in sorted(vals.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    if not any(s in k for s in ("rollout", "combine", "ilqr_", "mlp_jacobian", "mlp_forward")):
        continue
    lines.append(k[:120])
    for c in sorted(v):
        lines.append("    %-32s %.5g" % (c, v[c]))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE", 0) > 0:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over 256 CUs x 4 SIMDs
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        lines.append("    -> kernel cycles (per XCD)        %.5g" % cyc)
        lines.append("    -> MFMA busy fraction             %.3f" % (v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)))
    if "FETCH_SIZE" in v or "WRITE_SIZE" in v:
        # MI355X_MICROARCH.md, "HBM": both counters are in KB; on gfx950 FETCH_SIZE tallies the
        # 128-B requests of wide coalesced reads at 64 B, i.e. reports half the bytes -> doubled.
        # WRITE_SIZE is uncalibrated there and taken as reported.
        fetch = 2.0 * 1024.0 * v.get("FETCH_SIZE", 0)
        write = 1024.0 * v.get("WRITE_SIZE", 0)
        lines.append("    -> HBM bytes per launch: fetch %.5g (2 x FETCH_SIZE KB, gfx950 correction) + write %.5g = %.5g"
                     % (fetch, write, fetch + write))
        traffic[k] = {"fetch_bytes": fetch, "write_bytes": write, "bytes": fetch + write,
                      "fetch_size_kb_raw": v.get("FETCH_SIZE", 0), "write_size_kb_raw": v.get("WRITE_SIZE", 0)}
open(os.path.join(out, "rocprofv3_summary_%s.txt" % key), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
if traffic:
    path = os.path.join(out, "hbm_traffic.json")
    allt = json.load(open(path)) if os.path.exists(path) else {}
    allt[key] = {"command": cmd,
                 "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, average per launch; "
                           "fetch doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B)",
                 "kernels": traffic}
    json.dump(allt, open(path, "w"), indent=1)
