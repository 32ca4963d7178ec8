"""Condense the rocprofv3 CSVs of one tools/gpu_round.sh session into a small text summary
(per-kernel averages of every collected counter + derived MFMA-busy and HBM bytes)."""
import collections
import csv
import glob
import sys

out = sys.argv[1]
print("# rocprofv3 summary of `python bench.py --no-cpu-baseline` (c3, f64 headline + f32 side run), per launch averages")
for f in glob.glob(out + "/prof_trace/*/*_kernel_stats.csv"):
    print("\n## kernel-trace --stats")
    for row in csv.DictReader(open(f)):
        print("%-70s calls=%s avg_ns=%.0f pct=%s" % (row["Name"][:70], row["Calls"], float(row["AverageNs"]), row["Percentage"]))
vals = collections.defaultdict(dict)
for d in ("prof_pmc_sq", "prof_pmc_fetch", "prof_pmc_write", "prof_pmc_lds"):
    for f in glob.glob(out + "/" + d + "/*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            for c, x in v.items():
                vals[k][c] = sum(x) / len(x)
traffic = {}
print("\n## PMC (separate passes), average per launch")
for k, v in vals.items():
    if "rollout" not in k and "update" not in k:
        continue
    print(k[:90])
    for c in sorted(v):
        print("    %-32s %.4g" % (c, v[c]))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"] > 0:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over 256 CUs x 4 SIMDs
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        print("    -> kernel cycles (per XCD)        %.4g" % cyc)
        print("    -> MFMA busy fraction             %.3f" % (v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)))
    if "FETCH_SIZE" in v or "WRITE_SIZE" in v:
        # MI355X_MICROARCH.md, "HBM": both counters are in KB; on gfx950 FETCH_SIZE tallies the
        # 128-B requests of wide coalesced reads at 64 B, i.e. reports half the bytes -> doubled.
        # WRITE_SIZE is uncalibrated there and taken as reported.
        fetch = 2.0 * 1024.0 * v.get("FETCH_SIZE", 0)
        write = 1024.0 * v.get("WRITE_SIZE", 0)
        print("    -> HBM bytes per launch: fetch %.4g (2 x FETCH_SIZE KB, gfx950 correction) + write %.4g = %.4g"
              % (fetch, write, fetch + write))
        traffic[k] = {"fetch_bytes": fetch, "write_bytes": write, "bytes": fetch + write,
                      "fetch_size_kb_raw": v.get("FETCH_SIZE", 0), "write_size_kb_raw": v.get("WRITE_SIZE", 0)}
if traffic:
    import json
    json.dump({"command": "python bench.py --no-cpu-baseline (c3, f64, 1 solve per step)",
               "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, average per launch; "
                         "fetch doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B)",
               "kernels": traffic}, open(out + "/hbm_traffic.json", "w"), indent=1)
