#!/bin/bash
# ablation: the deterministic pass WITHOUT the fixed-point split of its weights (wrong sums, the same LDS adds): what a cheaper split could reach at most
mkdir -p gpurun_out/da
for lib in base abl base abl; do
  if [ $lib = abl ]; then export XC_LIB_PATH=$PWD/build/libxc_detabl.so (xc_hist_det.hip compiled with a three-bit-operation stand-in for det_split: see profiles/r06_notes.md, section 3); else unset XC_LIB_PATH; fi
  XC_BENCH_NO_SELFCHECK=1 timeout -k 10 150 python bench.py --deterministic --steps 30 --warmup 5 --no-cpu --no-extras --no-cfg4 > gpurun_out/da/o.json 2> gpurun_out/da/o.err
  echo "$lib rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/da/o.json').read().strip().splitlines()[-1])
    print('ms_per_step %.4f launch_ms %.4f' % (d['ms_per_step'], d['roofline'].get('launch_ms') or 0))
except Exception as e: print('no line', str(e)[:80], open('gpurun_out/da/o.err').read()[-200:].replace(chr(10),' '))
PY
)"
done
