#!/bin/bash
# tools/pmc_bench_traffic.sh (run on the GPU box via gpurun): fabric traffic of the bench's dominant kernel, FETCH_SIZE and
# WRITE_SIZE in SEPARATE --pmc passes (the guide's rule), for every schedule bench.py can run:
#   chain (default) | nochain, each with the shared dA plane and with per-slab dA (--slab-dA), and for float32 tracers (--dtype f32);
# plus rocprofv3 --kernel-trace --stats summaries of the default command and of --no-chain.
# The JSON carries sha256(xc_hist.hip + xc_hist_kernel.h + xc_binning.h): bench.py quotes a figure only while that matches its tree.
# Writes gpurun_out/hist_traffic.json (copy to profiles/) and gpurun_out/kt_<mode>/.
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && cd $R
COMMIT=${1:-unknown}
for mode in chain nochain slab_chain slab_nochain f32_chain f32_nochain det_chain; do
  arg=""
  case $mode in *nochain) arg="--no-chain";; esac
  case $mode in slab_*) arg="$arg --slab-dA";; esac
  case $mode in f32_*) arg="$arg --dtype f32";; esac
  case $mode in det_*) arg="$arg --deterministic";; esac
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmcb_${mode}_$c -- python3 bench.py --steps 12 --warmup 3 --no-cpu --no-extras --no-cfg4 $arg > /dev/null 2>&1
  done
done
for mode in chain nochain; do
  arg=""; [ $mode = nochain ] && arg="--no-chain"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$mode -o kt -- python3 bench.py --steps 100 --warmup 10 --no-cpu --no-extras --no-cfg4 $arg > $R/gpurun_out/kt_$mode.log 2>&1
done
python3 - <<PY
import csv,glob,collections,json,hashlib
B, NY, NX = 64, 1801, 3600
_h = hashlib.sha256()
for _f in ('xc_hist.hip', 'xc_hist_kernel.h', 'xc_binning.h'):
    _h.update(open('$R/xcontour_amd/csrc/' + _f, 'rb').read())
import datetime, subprocess, sys
sys.path.insert(0, '$R')
try:
    from xcontour_amd import _native as _nat
    _c = _nat.Context(0); _dev = _c.device_name(); _c.close()
except Exception as _e:
    _dev = 'unknown (%s)' % _e
out = {'commit': '$COMMIT', 'source_sha256': _h.hexdigest(), 'slabs_per_launch': B,
       'measured_at_utc': datetime.datetime.utcnow().strftime('%Y-%m-%dT%H:%M:%SZ'), 'device': _dev,
       'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over python3 bench.py --steps 12 --warmup 3 --no-cpu '
                 '[--no-chain] [--slab-dA]; median over the dispatches of the dominant kernel; FETCH_SIZE x 2 '
                 '(gfx950 reports half of a wide coalesced stream, MI355X_MICROARCH.md; calibrated on k_minmax_partial: '
                 'known bytes / reported KB) + WRITE_SIZE x 1; KB = 1024 B'}
raw = collections.defaultdict(dict)
for f in sorted(glob.glob("$R/gpurun_out/pmcb_*/*/*counter_collection.csv")):
    mode = f.split('/')[-3].replace('pmcb_', '').rsplit('_', 2)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        name = 'k_hist' if 'k_hist<' in k else ('k_minmax_partial' if 'k_minmax_partial' in k else None)
        if name:
            agg[(name, r['Counter_Name'])].append(float(r['Counter_Value']))
    for (name, ctr), v in agg.items():
        v = sorted(v); raw[mode][name + ':' + ctr] = v[len(v) // 2]
cal = None
if 'k_minmax_partial:FETCH_SIZE' in raw.get('nochain', {}):
    known = B * NY * NX * 8 / 1024.0
    cal = known / raw['nochain']['k_minmax_partial:FETCH_SIZE']
    out['fetch_calibration'] = {'kernel': 'k_minmax_partial<double>', 'known_KB': known, 'FETCH_SIZE_KB_raw': raw['nochain']['k_minmax_partial:FETCH_SIZE'], 'factor': cal}
for mode, d in raw.items():
    dom = 'k_hist'
    if dom + ':FETCH_SIZE' not in d: continue
    slab = mode.startswith('slab_')
    f32 = mode.startswith('f32_')
    cells = B * NY * NX
    out[mode] = {'kernel': dom, 'slabs_per_launch': B, 'commit': '$COMMIT', 'FETCH_SIZE_KB_raw': d[dom + ':FETCH_SIZE'], 'WRITE_SIZE_KB': d.get(dom + ':WRITE_SIZE'),
                 'hbm_bytes_per_launch': (d[dom + ':FETCH_SIZE'] * 2 + d.get(dom + ':WRITE_SIZE', 0.0)) * 1024,
                 'algorithmic_bytes_per_launch': cells * (12 if f32 else 16),
                 'hbm_unique_bytes_per_launch': cells * (4 if f32 else 8) + (cells * 8 if slab else NY * NX * 8)}
json.dump(out, open("$R/gpurun_out/hist_traffic.json", 'w'), indent=1)
print(json.dumps({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk in ('hbm_bytes_per_launch', 'FETCH_SIZE_KB_raw', 'factor')}) for k, v in out.items() if k != 'method'}, indent=1))
PY
