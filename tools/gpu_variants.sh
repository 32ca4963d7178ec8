#!/bin/bash
# On the GPU box (via gpurun): MFMA ceiling, per-phase breakdown, and the everything-but-MFMA time.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
variants/mfma_peak | tee gpurun_out/mfma_peak.log
python tools/phasetime.py f64 1 | tee gpurun_out/phasetime_f64.log
python tools/phasetime.py f32 1 | tee gpurun_out/phasetime_f32.log
for v in "" nomfma; do
  if [ -z "$v" ]; then unset AMPC_LIB; name=base; else export AMPC_LIB=$PWD/variants/lib_$v.so; name=$v; fi
  for prec in f64 f32; do
    timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 --precision $prec > gpurun_out/v_${name}_$prec.json 2> gpurun_out/v_${name}_$prec.err
    python -c "
import json
d=json.load(open('gpurun_out/v_${name}_$prec.json')); r=d['roofline']
print('%-10s %s kernel_ms=%.3f per-step-us=%.2f' % ('$name', '$prec', r['kernel_ms'], r['kernel_ms']*1e3/30))
"
  done
done
