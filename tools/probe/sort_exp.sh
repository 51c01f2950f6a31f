# kernel averages of the 6.48 M-pair sort for a list of builds: bash tools/probe/sort_exp.sh libA.so libB.so ...  (rocprofv3 --kernel-trace --stats)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out; cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export XC_LIB_PATH=$GRAFT_REPO_ROOT/xcontour_amd/$v
  rm -rf /tmp/kt_$v
  XC_REPS=${XC_REPS:-30} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -o t -- python3 $GRAFT_REPO_ROOT/tools/sort_only.py ${SHAPE:-1 1801 3600} > $GRAFT_REPO_ROOT/gpurun_out/exp_$v.log 2>&1
  echo "== $v"
  python3 - <<PY
import csv,glob
f=glob.glob('/tmp/kt_$v/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name'].replace('xc::(anonymous namespace)::','').replace('unsigned long long','u64')
    if 'scatter' in n or 'radix_hist' in n or 'fix_runs' in n or 'scan' in n: print(n[:60].ljust(60), r['Calls'].rjust(4), round(float(r['AverageNs'])/1e3,2))
PY
done
