#!/bin/bash
# tools/pmc_kernels.sh  (run on the GPU box via gpurun): HBM traffic of K7 / K8 / K9 from PMC counters,
# one counter per pass (FETCH_SIZE and WRITE_SIZE in separate runs, as the microarchitecture guide asks).
# Prints, per kernel, the median counter value per launch; FETCH_SIZE is in units of 32 B... see profiles/r01_notes.md
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  XC_CPU=0 XC_STRIDES=1 XC_SLABS=8 timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmck_cross_$c -- python3 tools/kernel_times.py cross > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmck_sort_$c -- python3 tools/kernel_times.py sort > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$R/gpurun_out/pmck_*/*/*counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r['Kernel_Name'][:70], r['Counter_Name'], r.get('Grid_Size',''))].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()):
        v=sorted(v); print(f.split('/')[-3], k, len(v), v[len(v)//2])
PY
