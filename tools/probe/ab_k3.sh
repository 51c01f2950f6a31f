# same-box timing of K3 variant builds (ablations compute WRONG sums: bench.py's self-check is expected to complain; only launch_ms is read):
#   bash tools/probe/ab_k3.sh "--dtype f32" libA.so libB.so ...
ARGS="$1"; shift
for v in "$@"; do
  XC_LIB_PATH=xcontour_amd/$v XC_BENCH_NO_SELFCHECK=1 python bench.py --no-cpu --no-extras --no-cfg4 --steps 30 --warmup 5 $ARGS 2>/dev/null | python -c "
import sys,json
ls=[l for l in sys.stdin if l.startswith('{')]
if ls:
    d=json.loads(ls[-1]); print('$v', '$ARGS', 'launch_ms', round(d['roofline']['launch_ms'],4), 'ms_per_step', round(d['ms_per_step'],4))
else: print('$v', 'no line')"
done
