#!/bin/bash
# 32 LDS copies of the histogram (an experiment build: build/libxc_c32.so, -DXC_MAX_COPIES=32 -DXC_LDS_BUDGET_KB=155) against the shipped 16
mkdir -p gpurun_out/c32
for lib in base c32 base c32; do
  for args in "--dtype f32" "--dtype f32 --variant 2" "--dtype f64" "--dtype f64 --no-chain" "--deterministic"; do
    if [ $lib = c32 ]; then export XC_LIB_PATH=$PWD/build/libxc_c32.so; else unset XC_LIB_PATH; fi
    timeout -k 10 150 python bench.py $args --steps 30 --warmup 5 --no-cpu --no-extras --no-cfg4 > gpurun_out/c32/o.json 2> gpurun_out/c32/o.err
    rc=$?
    echo "$lib [$args] rc=$rc $(python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/c32/o.json').read().strip().splitlines()[-1])
    print('ms_per_step %.4f launch_ms %.4f' % (d['ms_per_step'], d['roofline'].get('launch_ms') or 0))
except Exception as e: print('no line', e, open('gpurun_out/c32/o.err').read()[-300:])
PY
)"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  done
done
