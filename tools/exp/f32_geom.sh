#!/bin/bash
# float32 chained K3 (64 slabs per launch): launch geometry scan -- threads per workgroup x LDS copies x rows per wave
mkdir -p gpurun_out/geo
for cfg in "0 0 0" "512 0 0" "512 8 0" "768 0 0" "1024 8 0" "0 0 96" "0 0 384" "512 8 96" "512 8 384" "256 8 0" "256 4 0"; do
  set -- $cfg
  export XC_HIST_THREADS=$1 XC_HIST_NCOPY=$2 XC_HIST_ROWS=$3
  timeout -k 10 150 python bench.py --dtype f32 --steps 30 --warmup 5 --no-cpu --no-extras --no-cfg4 > gpurun_out/geo/g_$1_$2_$3.json 2> gpurun_out/geo/g.err
  rc=$?
  echo "threads $1 ncopy $2 rows $3 rc=$rc $(python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/geo/g_$1_$2_$3.json').read().strip().splitlines()[-1])
    print('ms_per_step %.4f launch_ms %.4f' % (d['ms_per_step'], d['roofline'].get('launch_ms') or 0))
except Exception as e: print('no line', e)
PY
)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
