#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 --kernel-trace run: per kernel name the average duration, and for the dominant kernel the
idle time between the end of one launch and the start of the next (what the epilogue / launch overhead costs per step).
usage: timeline_gaps.py <dir with *_kernel_trace.csv> [dominant-kernel substring, default k_hist]
       timeline_gaps.py <dir> --seq <kernel substring>    every launch between the last two launches of that kernel, with the idle
                                                          time in front of each (one call of a multi-kernel operator, e.g. the sort)"""
import csv
import glob
import sys

import numpy as np

d = sys.argv[1]
dom = sys.argv[2] if len(sys.argv) > 2 else 'k_hist'
rows = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', ''))))
rows.sort()
if len(sys.argv) > 3 and sys.argv[2] == '--seq':
    short = lambda n: n.replace('(anonymous namespace)::', '').replace('xc::', '').split('(')[0][:56]
    idx = [i for i, r in enumerate(rows) if sys.argv[3] in r[2]]
    if len(idx) < 2:
        sys.exit('fewer than two launches of ' + sys.argv[3])
    prev = None
    for s_, e_, n_, q_ in rows[idx[-2]:idx[-1] + 1]:
        print('%-58s %8.1f us   idle before %6.1f us' % (short(n_), (e_ - s_) / 1e3, (s_ - prev) / 1e3 if prev else 0.0))
        prev = e_
    print('start -> start: %.1f us' % ((rows[idx[-1]][0] - rows[idx[-2]][0]) / 1e3))
    sys.exit(0)
names = {}
for s, e, n, q in rows:
    names.setdefault(n.replace('(anonymous namespace)::', '').replace('xc::', '').split('(')[0][:60], []).append(e - s)
for n, v in sorted(names.items(), key=lambda kv: -sum(kv[1])):
    print('%-62s calls %5d  avg %10.1f us  total %10.1f us' % (n, len(v), np.mean(v) / 1e3, np.sum(v) / 1e3))
h = [(s, e) for s, e, n, q in rows if dom in n]
gaps = np.array([h[i + 1][0] - h[i][1] for i in range(len(h) - 1)]) / 1e3
per = np.array([h[i + 1][0] - h[i][0] for i in range(len(h) - 1)]) / 1e3
if len(gaps):
    print('%s: %d launches; end -> next start gap: median %.1f us, mean %.1f, p10 %.1f, p90 %.1f; start -> start median %.1f us'
          % (dom, len(h), np.median(gaps), gaps.mean(), np.percentile(gaps, 10), np.percentile(gaps, 90), np.median(per)))
# what else ran inside those gaps / under the dominant kernel
oth = [(s, e, n) for s, e, n, q in rows if dom not in n]
under = [min(e, he) - max(s, hs) for s, e, n in oth for hs, he in h if min(e, he) > max(s, hs)] if len(h) < 2000 else []
if under:
    print('time other kernels spent UNDER %s launches: %d overlaps, mean %.1f us' % (dom, len(under), np.mean(under) / 1e3))
