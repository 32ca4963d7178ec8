"""Rewrite the measurement table of DESIGN.md section 5 (between "## 5." and "## 6."; the history in Appendix H
is never touched) from the committed bench JSONs in profiles/, so that the table cannot drift from the evidence.
Usage: python tools/sync_design_table.py [tag]   (default r05)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"


def load(name):
    with open(os.path.join(ROOT, "profiles", "%s_bench_%s.json" % (TAG, name))) as f:
        return json.load(f)


def th(x):
    x = int(round(x))
    return "%d %03d" % (x // 1000, x % 1000) if x >= 10000 else str(x)


d=load('c3_f64'); b=load('c3_f64_batch8'); f32=load('c3_f32'); c1=load('c1_sindy_f64'); c2=load('c2_f64'); ax=load('arx_f64'); c4=load('c4_ilqr_f64'); c5=load('c5_candidates_f64'); drv=load('c3_f64_driver')
r=d['roofline']; sub=d.get('sub_records',{})
rw=d.get('repeat_windows',{})
out = []
out.append("| workload | precision | solves/s | dominant kernel | algorithmic TFLOP/s | of dense MFMA peak |\n|---|---|---|---|---|---|")
out.append("| **c3** HalfCheetah MPPI 4096×30 (headline) | f64 | **%s** | %.3f ms | %.1f | **%.1f %%** of 78.6 |"%(th(d['value']),r['kernel_ms'],r['achieved'],100*r['frac']))
fr=f32['roofline']
out.append("| c3 | f32 | %s | %.3f ms | %.1f | %.1f %% of 157.3 |"%(th(f32['value']),fr['kernel_ms'],fr['achieved'],100*fr['frac']))
rb=b['roofline']
out.append("| c3, 8 independent solves per launch (64-row tiles) | f64 | %s | %.2f ms | %.1f | %.1f %% |"%(th(b['value']),rb['kernel_ms'],rb['achieved'],100*rb['frac']))
out.append("| c2 Pendulum MPPI 1024×30 (four-row kernel §4.1b; latency-bound) | f64 | %s | %.3f ms | %.1f | %.1f %% |"%(th(c2['value']),c2['roofline']['kernel_ms'],c2['roofline']['achieved'],100*c2['roofline']['frac']))
out.append("| arx: MPPI 1024×30 on a 20-state ARX model (latency-bound) | f64 | %s | %.3f ms | %.1f | — |"%(th(ax['value']),ax['roofline']['kernel_ms'],ax['roofline']['achieved']))
out.append("| c1 CartPole SINDy MPPI 256×20 (§4.5; latency-bound) | f64 | %s | %.3f ms | — | — |"%(th(c1['value']),c1['roofline']['kernel_ms']))
k4=c4['roofline']
def subv(rec, key):
    return th(rec[key]['value']) if isinstance(rec.get(key), dict) else "—"


out.append("| **c4** HalfCheetah iLQR H=50, converging set, %d problems %s | f64 | **%s** (the same problems streamed through 256 slots: %s; 4096 problems at once: %s; 4096 through %d slots: %s; lock-step batches of 256: %s) | %s %.3f ms per launch | %.1f (whole solve) | %.0f %% |"
           % (c4['problems_per_step'], "admitted at once (%d slots)" % c4['slots'] if c4['slots'] >= c4['problems_per_step'] else "through %d slots" % c4['slots'],
              th(c4['value']), subv(c4, 'slots_256'), subv(c4, 'all_at_once_4096'), c4['slots'], subv(c4, 'stream_4096'), subv(c4, 'lockstep_batches'),
              k4['kernel'], k4['kernel_ms'], c4['algorithmic_tflops'], 100 * c4['algorithmic_tflops'] / 78.6))
out.append("| c5 64 candidates × 200-row closed loop, scored on device | f64 | %s (MPPI solves) | rollout %.3f ms per control step | %.1f (whole closed loop) | %.1f %% |"%(th(c5['value']),c5['roofline']['kernel_ms'],c5['algorithmic_tflops'],100*c5['algorithmic_tflops']/78.6))
ie=sub.get('ilqr_eval')
if ie: out.append("| iLQR candidates: 64 × 49 control steps, horizons 5–25 in one plan (`sub_records.ilqr_eval`) | f64 | %s (full solves) | — | — | — |"%th(ie['value']))
c4b = load('c4_ilqr_f64_b256')
rows = [l for l in out if l.startswith("| ") and not l.startswith("| workload")]
rows.insert(7, "| c4, 1024 problems through 256 slots (rounds 3–4) | f64 | %d | — | %.1f | %.0f %% |"
            % (round(c4b['value']), c4b['algorithmic_tflops'], 100 * c4b['algorithmic_tflops'] / 78.6))
path = os.path.join(ROOT, "DESIGN.md")
lines = open(path).read().split("\n")
lo = next(i for i, l in enumerate(lines) if l.startswith("## 5."))
hi = next(i for i, l in enumerate(lines) if l.startswith("## 6."))
first = next(i for i in range(lo, hi) if lines[i].startswith("|---"))
last = first
while last + 1 < hi and lines[last + 1].startswith("|"):
    last += 1
lines[first + 1:last + 1] = rows
open(path, "w").write("\n".join(lines))
print("\n".join(rows))
