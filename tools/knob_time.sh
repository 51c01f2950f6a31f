#!/bin/bash
# tools/knob_time.sh "<ENV=VAL ...>" <tag> [bench args]: launch time of the bench's histogram kernel under experiment knobs
cd $GRAFT_REPO_ROOT
for kv in $1; do export $kv; done
tag=$2; shift 2
mkdir -p gpurun_out/knob
python3 bench.py --steps 60 --warmup 5 --no-cpu "$@" > gpurun_out/knob/$tag.json 2> gpurun_out/knob/$tag.err
python3 -c "
import json; b = json.load(open('gpurun_out/knob/$tag.json'))
print('$tag', 'launch_ms', round(b['roofline']['launch_ms'], 4), 'value %.3e' % b['value'], 'nochain %.3e' % b['unchained']['value'], flush=True)"
