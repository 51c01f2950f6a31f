#!/bin/bash
# tools/ab.sh libA.so libB.so [bench args]: same-box A/B of two builds of the library (tools/build_variant.sh), alternating runs
cd $GRAFT_REPO_ROOT
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for L in $A $B; do
    XC_LIB_PATH=$GRAFT_REPO_ROOT/xcontour_amd/$L python3 bench.py --steps 60 --warmup 5 --no-cpu "$@" 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('$L', 'launch_ms %.4f' % b['roofline']['launch_ms'], 'value %.3e' % b['value'], 'nochain %.3e' % b.get('unchained',{}).get('value',0), flush=True)"
  done
done
