#!/bin/bash
# the 925 KB result of cal_local_wave_activity on one 241 x 480 plane: DMA copy (XC_COPY_OUT_KB=256) against the copy kernel (1024)
for kb in 256 1024 256 1024; do
  export XC_COPY_OUT_KB=$kb
  timeout -k 10 100 python3 tools/exp/lwa_facade.py > gpurun_out/lwa_kb.json || exit 1
  python3 - <<'PY'
import json, os
d = json.load(open('gpurun_out/lwa_kb.json'))
print('XC_COPY_OUT_KB', os.environ['XC_COPY_OUT_KB'], {k: v['us'] for k, v in d.items()})
PY
done
