#!/bin/bash
# tools/pmc_k3.sh [bench args]: instruction mix of the bench's histogram kernels (chained and unchained variants separately)
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && cd $R
rm -rf $R/gpurun_out/pmck3_*
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_BRANCH" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmck3_$i -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-extras --no-cfg4 "$@" > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections
cells = 64 * 1801 * 3600
for f in sorted(glob.glob("$R/gpurun_out/pmck3_*/*/*counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_hist<' in k:
            agg[(('f32 ' if 'k_hist<float' in k else '') + ('chain' if 'true, true, true, true' in k else 'nochain'), r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()):
        v=sorted(v); m=v[len(v)//2]; print(k[0], k[1], '%.4g' % m, 'per cell %.4f' % (m / cells))
PY
