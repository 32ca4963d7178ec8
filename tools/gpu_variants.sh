#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
variants/mfma_peak | tee gpurun_out/mfma_peak.log
for v in "" nocost noload nomfma nomfma_noload; do
  if [ -z "$v" ]; then unset AMPC_LIB; name=base; else export AMPC_LIB=$PWD/variants/lib_$v.so; name=$v; fi
  for prec in f64 f32; do
    timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 10 --precision $prec > gpurun_out/v_${name}_$prec.json 2> gpurun_out/v_${name}_$prec.err
    python -c "
import json
d=json.load(open('gpurun_out/v_${name}_$prec.json')); r=d['roofline']
print('%-16s %s kernel_ms=%.3f per-step-us=%.2f' % ('$name', '$prec', r['kernel_ms'], r['kernel_ms']*1e3/30))
"
  done
done
