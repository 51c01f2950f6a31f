#!/bin/bash
# tools/collect_profiles.sh ROUND COMMIT [a|b]  (run on the GPU box via gpurun; part a ~8 min: bench lines, rehearsals, timings; part b ~10 min: rocprofv3
# stats + PMC + the default bench line; no third argument: both): everything profiles/ holds for a round, re-taken at one commit.  Writes gpurun_out/rNN_*; copy what
# should be judged into profiles/.
R=$GRAFT_REPO_ROOT; cd $R; RD=${1:-r05}; COMMIT=${2:-unknown}; PART=${3:-ab}; O=$R/gpurun_out; mkdir -p $O
case $PART in *a*)
echo "== bench variants"
: > $O/${RD}_bench_variants.jsonl
for a in "--no-chain" "--slab-dA" "--row-dA" "--deterministic" "--variant 1" "--variant 2" "--variant 3" "--dtype f32" "--dtype f32 --no-chain" "--dtype f32 --variant 3"; do
  timeout -k 10 300 python3 bench.py --no-cpu --no-cfg4 --no-extras $a 2>/dev/null | grep '^{' >> $O/${RD}_bench_variants.jsonl || exit 1
done
echo "== secondary configs"
for c in cfg3 cfg4 cfg5; do timeout -k 10 300 python3 bench.py --config $c --steps 50 --warmup 5 2>/dev/null | grep '^{' > $O/${RD}_bench_$c.json || exit 1; done
echo "== 2 and 4 ranks on this one GPU, the driver's own command form (the N > 1 path of bench.py incl. cfg4_strong: RCCL refuses a shared GPU, the ladder lands on HIP IPC; correctness + gather evidence, not a speed)"
timeout -k 10 500 python3 bench.py --gpus 2 --cpu-slabs 2 2> $O/${RD}_bench_2ranks_ipc_1gpu.err | grep '^{' > $O/${RD}_bench_2ranks_ipc_1gpu.json || exit 1
timeout -k 10 500 python3 bench.py --gpus 4 --steps 6 --batch 16 --cfg4-slabs 4736 --cpu-slabs 2 2> $O/${RD}_bench_4ranks_ipc_1gpu.err | grep '^{' > $O/${RD}_bench_4ranks_ipc_1gpu.json || exit 1
echo "== kernel timings"
timeout -k 10 500 python3 tools/kernel_times.py sort lwa cross pipe land single > $O/${RD}_kernel_times.jsonl 2>&1 || exit 1
XC_FACADE_STACK=128 XC_FACADE_SMALL=1 timeout -k 10 300 python3 tools/facade_time.py 2>/dev/null | grep '^{' > $O/${RD}_facade_time.jsonl || exit 1
XC_FACADE_KW='{"resident": true}' XC_FACADE_SMALL=1 timeout -k 10 300 python3 tools/facade_time.py 2>/dev/null | grep '^{' >> $O/${RD}_facade_time.jsonl || exit 1
XC_FACADE_SMALL=1 timeout -k 10 300 python3 tools/facade_time.py --breakdown 2>/dev/null | grep '^{' | tail -1 >> $O/${RD}_facade_time.jsonl || exit 1
XC_FACADE_KW='{"resident": true}' XC_FACADE_SMALL=1 timeout -k 10 300 python3 tools/facade_time.py --breakdown 2>/dev/null | grep '^{' | tail -1 >> $O/${RD}_facade_time.jsonl || exit 1
;; esac
case $PART in *b*)
echo "== rocprofv3 kernel stats + PMC"
cd /tmp && export TMPDIR=/tmp && cd $R
bash tools/pmc_bench_traffic.sh $COMMIT > $O/${RD}_pmc_traffic.log 2>&1 || exit 1
cp $O/kt_chain/*/kt_kernel_stats.csv $O/${RD}_bench_kernel_stats_chain.csv 2>/dev/null || cp $O/kt_chain/kt_kernel_stats.csv $O/${RD}_bench_kernel_stats_chain.csv
cp $O/kt_nochain/*/kt_kernel_stats.csv $O/${RD}_bench_kernel_stats_nochain.csv 2>/dev/null || cp $O/kt_nochain/kt_kernel_stats.csv $O/${RD}_bench_kernel_stats_nochain.csv
bash tools/pmc_k3.sh > $O/${RD}_pmc_k3_instruction_mix.txt 2>&1 || exit 1
bash tools/pmc_k3.sh --deterministic > $O/${RD}_pmc_k3_deterministic_instruction_mix.txt 2>&1 || exit 1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_single -o kt -- python3 tools/single_slab.py 40 > /dev/null 2>&1
cp $O/kt_single/*/kt_kernel_stats.csv $O/${RD}_single_slab_kernel_stats.csv 2>/dev/null || cp $O/kt_single/kt_kernel_stats.csv $O/${RD}_single_slab_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_det -o kt -- python3 bench.py --deterministic --steps 100 --warmup 10 --no-cpu --no-extras --no-cfg4 > /dev/null 2>&1
cp $O/kt_det/*/kt_kernel_stats.csv $O/${RD}_bench_kernel_stats_deterministic.csv 2>/dev/null || cp $O/kt_det/kt_kernel_stats.csv $O/${RD}_bench_kernel_stats_deterministic.csv
bash tools/pmc_lwa.sh > $O/${RD}_pmc_k7_instruction_mix.txt 2>&1 || exit 1
bash tools/pmc_cross.sh > $O/${RD}_pmc_k9_instruction_mix.txt 2>&1 || exit 1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_sort -o kt -- python3 tools/kernel_times.py sort > /dev/null 2>&1
cp $O/kt_sort/*/kt_kernel_stats.csv $O/${RD}_k8_sort_kernel_stats.csv 2>/dev/null || cp $O/kt_sort/kt_kernel_stats.csv $O/${RD}_k8_sort_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg4 -o kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras > /dev/null 2>&1
cp $O/kt_cfg4/*/kt_kernel_stats.csv $O/${RD}_bench_cfg4_strong_kernel_stats.csv 2>/dev/null || cp $O/kt_cfg4/kt_kernel_stats.csv $O/${RD}_bench_cfg4_strong_kernel_stats.csv
echo "== bench default (last: it quotes the PMC traffic just measured)"
cp $O/hist_traffic.json $R/profiles/hist_traffic.json
timeout -k 10 500 python3 bench.py > $O/${RD}_bench_n1.json 2> $O/${RD}_bench_n1.err || exit 1
;; esac
echo "== done"; ls -la $O/${RD}_*
