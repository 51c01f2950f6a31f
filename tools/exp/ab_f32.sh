#!/bin/bash
# ablation: float32 K3 with the stencil in float32 (build/libxc_gf32.so) against the shipped library; prints launch_ms per field
mkdir -p gpurun_out/ab
for lib in base gf32; do
  for v in 0 2 3; do
    if [ $lib = gf32 ]; then export XC_LIB_PATH=$PWD/build/libxc_gf32.so; else unset XC_LIB_PATH; fi
    timeout -k 10 150 python bench.py --dtype f32 --variant $v --steps 30 --warmup 5 --no-cpu --no-extras --no-cfg4 > gpurun_out/ab/${lib}_v$v.json 2> gpurun_out/ab/${lib}_v$v.err
    rc=$?
    echo "$lib v$v rc=$rc $(python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/ab/${lib}_v$v.json').read().strip().splitlines()[-1])
    print('ms_per_step', d['ms_per_step'], 'launch_ms', d.get('roofline',{}).get('launch_ms'), 'parity', d.get('parity'))
except Exception as e: print('no line', e)
PY
)"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  done
done
