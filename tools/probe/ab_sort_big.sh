# same-box A/B of builds on the 6.48 M-pair sorts: bash tools/probe/ab_sort_big.sh libA.so libB.so ...
for i in 1 2; do for v in "$@"; do
  XC_LIB_PATH=xcontour_amd/$v XC_SORT_ONLY=1 python tools/kernel_times.py sort 2>&1 | python -c "
import sys,json
print('$v', [(json.loads(l)['dtype'], round(json.loads(l)['ms'],4)) for l in sys.stdin if l.startswith('{')])"
done; done
