#!/bin/bash
# the reference's call sequence at its demo size, resident inputs, with the library's breakdown: XC_COPY_KERNEL 0 against 1
export XC_FACADE_SMALL=1 XC_FACADE_KW='{"resident": true}'
for ck in 0 1 0 1; do
  XC_COPY_KERNEL=$ck timeout -k 10 200 python3 tools/facade_time.py --breakdown > gpurun_out/facade_ck${ck}.jsonl 2>&1 || exit 1
  python3 - $ck <<'PY'
import json, sys
for l in open('gpurun_out/facade_ck%s.jsonl' % sys.argv[1]):
    try: d = json.loads(l)
    except Exception: continue
    k = 'facade_us_per_call_cfg1_stack_15x241x480_f32'
    if k in d:
        r = d[k]
        print('XC_COPY_KERNEL', sys.argv[1], 'resident', d.get('resident'))
        for n, v in r.items():
            print('   %-32s us %6.1f  py %5.1f lib %5.1f wait %5.1f launch %5.1f' % (n, v['us'], v['python_us'], v['library_us'], v['library_split_us']['wait_for_stream'], v['library_split_us']['checks_and_launches']))
        print('  seq sum', round(sum(v['us'] for n, v in r.items() if not n.startswith('keff')), 1))
PY
done
