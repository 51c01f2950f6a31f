#!/bin/bash
cd /tmp && export TMPDIR=/tmp XC_LOOP=hist
R=$GRAFT_REPO_ROOT
for cfg in "0 0" "512 0" "256 0" "1024 4" "512 8" "256 16" "256 32" "128 32"; do
  set -- $cfg
  export XC_HIST_THREADS=$1 XC_HIST_BPS=$2
  rm -rf $R/gpurun_out/ft_$1_$2
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ft_$1_$2 -- python3 $R/tools/exp/facade_loop.py > $R/gpurun_out/ft.log 2>&1 || exit 1
  echo "== threads $1 bps $2"; python3 $R/tools/exp/trace_tail.py $R/gpurun_out/ft_$1_$2 | grep median
done
