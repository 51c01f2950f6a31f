mkdir -p $GRAFT_REPO_ROOT/gpurun_out; cd /tmp && export TMPDIR=/tmp
for v in xcontour_hip xc_norank xc_nostore xc_neither; do
  export XC_LIB_PATH=$GRAFT_REPO_ROOT/xcontour_amd/lib$v.so
  rm -rf /tmp/kt_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -o t -- python3 $GRAFT_REPO_ROOT/tools/sort_only.py > $GRAFT_REPO_ROOT/gpurun_out/exp_$v.log 2>&1
  echo "== $v"
  python3 - <<PY
import csv,glob
f=glob.glob('/tmp/kt_$v/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'scatter' in n or 'radix_hist' in n or 'fix_runs' in n: print(n[28:90], r['Calls'], round(float(r['AverageNs'])/1e3,2))
PY
done
