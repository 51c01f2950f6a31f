#!/bin/bash
# tools/pmc_lwa.sh (run on the GPU box via gpurun): instruction mix of K7 (k_lwa) on cfg3 and on a 64-slab stack of it:
# is it VALU-bound?  SQ_INSTS_VALU / SQ_BUSY_CYCLES etc. in their own --pmc passes (no trace domains alongside).
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && cd $R
rm -rf $R/gpurun_out/pmclwa_*
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_BRANCH SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmclwa_$i -- python3 tools/kernel_times.py lwa > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$R/gpurun_out/pmclwa_*/*/*counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_lwa<' in k:
            agg[(k.split('k_lwa')[1][:22], r.get('Grid_Size', r.get('Grid_Size_X','')), r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()):
        v=sorted(v); print('k_lwa%s grid %s %-24s %.4g' % (k[0], k[1], k[2], v[len(v)//2]))
PY
