cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && cd $R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_cfg4 -o kt -- python3 tools/gpu_persist_check.py --time-only --cfg4 --slabs 256 > $R/gpurun_out/kt_cfg4.log 2>&1
tail -4 $R/gpurun_out/kt_cfg4.log
python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/kt_cfg4/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:90], r['Calls'], r['AverageNs'], r['Percentage'])
PY
