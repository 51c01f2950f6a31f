#!/bin/bash
# tools/pmc_bench_traffic.sh (run on the GPU box via gpurun): HBM traffic of the bench's kernels, FETCH_SIZE and
# WRITE_SIZE in separate --pmc passes, for the default (chained) schedule and for --no-chain; plus the
# kernel-trace statistics of the default command.  Summaries go to gpurun_out/.
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && cd $R
for mode in chain nochain; do
  arg=""; [ $mode = nochain ] && arg="--no-chain"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmcb_${mode}_$c -- python3 bench.py --steps 20 --warmup 5 --no-cpu $arg > /dev/null 2>&1
  done
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$mode -o kt -- python3 bench.py --steps 100 --warmup 10 --no-cpu $arg > $R/gpurun_out/kt_$mode.log 2>&1
done
python3 - <<PY
import csv,glob,collections,json
out={}
for f in sorted(glob.glob("$R/gpurun_out/pmcb_*/*/*counter_collection.csv")):
    mode=f.split('/')[-3].replace('pmcb_','')
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_hist' in r['Kernel_Name'] or 'k_minmax_partial' in r['Kernel_Name']:
            agg[(r['Kernel_Name'].split('(')[0].replace('void xc::(anonymous namespace)::',''), r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()):
        v=sorted(v); out['%s | %s | %s' % (mode, k[0], k[1])] = v[len(v)//2]
print(json.dumps(out, indent=1))
PY
