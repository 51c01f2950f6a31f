#!/bin/bash
# tools/pmc_cross.sh (on the GPU box): instruction mix of K9 (k_crossing, stride 1) on the noisy PV-like cfg2 field and on the smooth one:
# what holds the noisy case at 4.5 TB/s?  SQ counters in their own --pmc passes.
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && cd $R
rm -rf $R/gpurun_out/pmccross_*
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  XC_CPU=0 XC_STRIDES=1 XC_SLABS=8 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmccross_$i -- python3 tools/kernel_times.py cross > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$R/gpurun_out/pmccross_*/*/*counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_crossing' in k and 'reduce' not in k:
            agg[(k.replace('xc::(anonymous namespace)::','')[:60], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()):
        v=sorted(v); print('%-62s %-24s n=%3d median %.4g  min %.4g max %.4g' % (k[0], k[1], len(v), v[len(v)//2], v[0], v[-1]))
PY
