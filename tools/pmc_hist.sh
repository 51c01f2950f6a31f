#!/bin/bash
# tools/pmc_hist.sh  (run on the GPU box via gpurun): PMC passes for k_hist at the bench default
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && cd $R
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmch_$i -- python3 bench.py --steps 6 --warmup 2 --no-cpu $BENCH_ARGS > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$R/gpurun_out/pmch_*/*/*counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_hist' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()):
        v=sorted(v); print(k, v[len(v)//2])
PY
