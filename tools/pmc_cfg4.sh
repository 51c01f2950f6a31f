#!/bin/bash
# tools/pmc_cfg4.sh (on the GPU box): fabric traffic of the histogram pass on cfg4-sized launch sets (256 slabs of 1440 x 721),
# FETCH_SIZE x 2 + WRITE_SIZE from separate --pmc passes over bench.py's cfg4_strong leg (a shortened stack).
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc4_$c -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras --cfg4-slabs 2048 --cfg4-reps 1 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections, json
v = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmc4_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'k_hist<' in r['Kernel_Name'] and int(r['Grid_Size']) == 768 * 1024:
            v[r['Counter_Name']].append(float(r['Counter_Value']))
med = {k: sorted(x)[len(x) // 2] for k, x in v.items()}
cells = 256 * 721 * 1440
b = (med.get('FETCH_SIZE', 0) * 2 + med.get('WRITE_SIZE', 0)) * 1024
print(json.dumps({'cfg4_launch_set': {'slabs': 256, 'dispatches': {k: len(x) for k, x in v.items()}, 'median_KB_raw': med, 'fabric_bytes': b,
                                      'algorithmic_bytes': cells * 16, 'streamed_bytes_chained': cells * 24, 'bytes_per_cell': b / cells}}))
PY
