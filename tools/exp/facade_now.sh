#!/bin/bash
# the reference's call sequence at its demo size, resident inputs, with the library's breakdown (three runs)
export XC_FACADE_SMALL=1 XC_FACADE_KW='{"resident": true}'
for i in 1 2 3; do
  timeout -k 10 200 python3 tools/facade_time.py --breakdown > gpurun_out/facade_now$i.jsonl 2>&1 || exit 1
  python3 - $i <<'PY'
import json, sys
for l in open('gpurun_out/facade_now%s.jsonl' % sys.argv[1]):
    try: d = json.loads(l)
    except Exception: continue
    k = 'facade_us_per_call_cfg1_stack_15x241x480_f32'
    if k in d:
        r = d[k]
        for n, v in r.items():
            if not isinstance(v, dict):
                print('   %-32s us %6.1f' % (n, v)); continue
            print('   %-32s us %6.1f  py %5.1f lib %5.1f wait %5.1f launch %5.1f' % (n, v['us'], v['python_us'], v['library_us'], v['library_split_us']['wait_for_stream'], v['library_split_us']['checks_and_launches']))
        print('  seq sum', round(sum(v['us'] for n, v in r.items() if isinstance(v, dict) and not n.startswith('keff')), 1))
PY
done
