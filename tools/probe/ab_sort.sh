# same-box A/B of two builds on the K8 small-plane cases: bash tools/probe/ab_sort.sh libA.so libB.so
for i in 1 2 3; do for v in "$@"; do
  XC_LIB_PATH=xcontour_amd/$v python bench.py --config cfg5 --steps 300 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v cfg5', round(d['ms_per_step'],4))"
  XC_LIB_PATH=xcontour_amd/$v python tools/kernel_times.py sort 2>&1 | grep batch | python -c "
import sys,json
print('$v', [ (json.loads(l)['planes'], round(json.loads(l)['ms'],4)) for l in sys.stdin])"
done; done
