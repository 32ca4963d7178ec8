"""Copy the results of one `tools/gpu_round.sh <tag>` session from gpurun_out/<tag>/ into profiles/
as <tag>_<name> (tracked): bench JSON lines (only the JSON line of each file), logs, rocprofv3
summaries, kernel-stats CSVs, the HBM traffic record.

    python tools/collect_profiles.py r02
"""
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
for f in sorted(glob.glob(os.path.join(src, "*"))):
    name = os.path.basename(f)
    if os.path.isdir(f) or name.endswith(".err") or name.startswith("profile_"):
        continue
    out = os.path.join(dst, "%s_%s" % (tag, name))
    if name.startswith("bench_") and name.endswith(".json"):
        line = None
        for ln in open(f):
            ln = ln.strip()
            if ln.startswith("{"):
                try:
                    json.loads(ln)
                    line = ln
                except ValueError:
                    pass
        if line is None:
            print("no JSON line in", name)
            continue
        open(out, "w").write(line + "\n")
    else:
        shutil.copy(f, out)
    print("profiles/%s_%s" % (tag, name))
