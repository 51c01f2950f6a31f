import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
short = lambda n: n.replace('(anonymous namespace)::', '').replace('xc::', '').split('(')[0][:70]
# the last 3 calls of each phase: find phase boundaries by kernel name sets
prev = None
out = []
for s, e, n in rows:
    out.append('%-72s %7.1f us  idle before %6.1f' % (short(n), (e - s) / 1e3, (s - prev) / 1e3 if prev else 0))
    prev = e
N = len(out)
for frac in (0.2, 0.45, 0.7, 0.97):
    i = int(N * frac)
    print('---- around launch', i)
    print('\n'.join(out[i:i + 10]))

import collections
agg = collections.defaultdict(list)
for s_, e_, n_ in rows[len(rows) // 2:]:
    agg[short(n_)].append((e_ - s_) / 1e3)
for k, v in agg.items():
    v.sort(); print('%-72s n %4d  median %6.1f us' % (k, len(v), v[len(v) // 2]))
