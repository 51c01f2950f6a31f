mkdir -p gpurun_out/r05
for lib in default nomerge merge2; do
  for dt in f32 f64; do
    for v in 0 2 3; do
      if [ $lib = default ]; then L=""; else L="XC_LIB_PATH=xcontour_amd/libxc_$lib.so"; fi
      r=$(env $L python bench.py --dtype $dt --variant $v --no-cpu --no-cfg4 --no-extras --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms launch %.4f value %.3e' % (d['ms_per_step'], d['roofline']['launch_ms'], d['value']))")
      echo "$lib $dt v$v $r"
    done
  done
done
