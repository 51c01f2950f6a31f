#!/bin/bash
# tools/pmc_knob.sh "<ENV=VAL ...>" <tag> [bench args]: fabric traffic (FETCH_SIZE, separate pass) + launch time of the
# bench's histogram kernel under experiment knobs (XC_HIST_BPS, XC_HIST_XCDMAP, XC_HIST_ROWS, ...).  Run on the GPU box.
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && cd $R
for kv in $1; do export $kv; done
tag=$2; shift 2
mkdir -p $R/gpurun_out/knob
python3 bench.py --steps 40 --warmup 5 --no-cpu "$@" > $R/gpurun_out/knob/$tag.json 2> $R/gpurun_out/knob/$tag.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/knob/pmc_$tag -- python3 bench.py --steps 8 --warmup 2 --no-cpu "$@" > /dev/null 2>&1
python3 - <<PY
import csv, glob, json
v = []
for f in glob.glob("$R/gpurun_out/knob/pmc_$tag/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'k_hist<' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
            v.append(float(r['Counter_Value']))
v.sort()
b = json.load(open("$R/gpurun_out/knob/$tag.json"))
print("$tag", "launch_ms", round(b['roofline']['launch_ms'], 4), "value", "%.3e" % b['value'],
      "fetch_GB", round(v[len(v) // 2] * 2 * 1024 / 1e9, 3) if v else None, flush=True)
PY
