#!/bin/bash
# tools/pmc_persist.sh (GPU box): FETCH_SIZE / WRITE_SIZE of the Keff kernels, persistent vs two-pass, plane vs per-slab dA
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmcp_$c -- python3 tools/gpu_persist_check.py --time-only --slabs ${SLABS:-16} > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$R/gpurun_out/pmcp_*/*/*counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void xc::(anonymous namespace)::','')
        if 'k_keff_persist' in k or 'k_hist' in k or 'k_minmax_partial' in k:
            agg[(k, r['Counter_Name'], r.get('Grid_Size',''))].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()):
        v=sorted(v); print(k, 'n=%d median %.1f KB' % (len(v), v[len(v)//2]))
PY
