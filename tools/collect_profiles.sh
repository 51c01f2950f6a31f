#!/bin/bash
# tools/collect_profiles.sh ROUND COMMIT  (run on the GPU box via gpurun, ~10 min): everything profiles/ holds for a round,
# re-taken at one commit.  Writes gpurun_out/rNN_*; copy what should be judged into profiles/.
R=$GRAFT_REPO_ROOT; cd $R; RD=${1:-r04}; COMMIT=${2:-unknown}; O=$R/gpurun_out
echo "== bench variants"
: > $O/${RD}_bench_variants.jsonl
for a in "--no-chain" "--slab-dA" "--row-dA" "--deterministic" "--variant 1" "--variant 2" "--dtype f32" "--dtype f32 --no-chain"; do
  timeout -k 10 300 python3 bench.py --no-cpu --no-cfg4 --no-extras $a 2>/dev/null | grep '^{' >> $O/${RD}_bench_variants.jsonl || exit 1
done
echo "== secondary configs"
for c in cfg3 cfg4 cfg5; do timeout -k 10 300 python3 bench.py --config $c --steps 50 --warmup 5 2>/dev/null | grep '^{' > $O/${RD}_bench_$c.json || exit 1; done
echo "== 2 ranks on this one GPU, gloo (the N > 1 path of bench.py incl. cfg4_strong; correctness evidence, not a speed)"
timeout -k 10 500 python3 bench.py --gpus 2 --backend gloo --steps 20 --warmup 3 2>/dev/null | grep '^{' > $O/${RD}_bench_2ranks_gloo_1gpu.json || exit 1
echo "== kernel timings"
timeout -k 10 500 python3 tools/kernel_times.py sort lwa cross pipe land single > $O/${RD}_kernel_times.jsonl 2>&1 || exit 1
XC_FACADE_STACK=128 XC_FACADE_SMALL=1 timeout -k 10 300 python3 tools/facade_time.py 2>/dev/null | grep '^{' > $O/${RD}_facade_time.jsonl || exit 1
XC_FACADE_KW='{"resident": true}' XC_FACADE_SMALL=1 timeout -k 10 300 python3 tools/facade_time.py 2>/dev/null | grep '^{' >> $O/${RD}_facade_time.jsonl || exit 1
echo "== rocprofv3 kernel stats + PMC"
cd /tmp && export TMPDIR=/tmp && cd $R
bash tools/pmc_bench_traffic.sh $COMMIT > $O/${RD}_pmc_traffic.log 2>&1 || exit 1
cp $O/kt_chain/*/kt_kernel_stats.csv $O/${RD}_bench_kernel_stats_chain.csv 2>/dev/null || cp $O/kt_chain/kt_kernel_stats.csv $O/${RD}_bench_kernel_stats_chain.csv
cp $O/kt_nochain/*/kt_kernel_stats.csv $O/${RD}_bench_kernel_stats_nochain.csv 2>/dev/null || cp $O/kt_nochain/kt_kernel_stats.csv $O/${RD}_bench_kernel_stats_nochain.csv
bash tools/pmc_k3.sh > $O/${RD}_pmc_k3_instruction_mix.txt 2>&1 || exit 1
bash tools/pmc_lwa.sh > $O/${RD}_pmc_k7_instruction_mix.txt 2>&1 || exit 1
bash tools/pmc_cross.sh > $O/${RD}_pmc_k9_instruction_mix.txt 2>&1 || exit 1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_sort -o kt -- python3 tools/kernel_times.py sort > /dev/null 2>&1
cp $O/kt_sort/*/kt_kernel_stats.csv $O/${RD}_k8_sort_kernel_stats.csv 2>/dev/null || cp $O/kt_sort/kt_kernel_stats.csv $O/${RD}_k8_sort_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg4 -o kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras > /dev/null 2>&1
cp $O/kt_cfg4/*/kt_kernel_stats.csv $O/${RD}_bench_cfg4_strong_kernel_stats.csv 2>/dev/null || cp $O/kt_cfg4/kt_kernel_stats.csv $O/${RD}_bench_cfg4_strong_kernel_stats.csv
echo "== bench default (last: it quotes the PMC traffic just measured)"
cp $O/hist_traffic.json $R/profiles/hist_traffic.json
timeout -k 10 500 python3 bench.py > $O/${RD}_bench_n1.json 2> $O/${RD}_bench_n1.err || exit 1
echo "== done"; ls -la $O/${RD}_*
