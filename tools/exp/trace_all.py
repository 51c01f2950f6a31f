import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('xc::', '').split('(')[0][:60]))
for f in glob.glob(sys.argv[1] + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C %s %s' % (r.get('Direction', ''), r.get('Size', r.get('Bytes', '')))))
rows.sort()
n = len(rows); i0 = int(n * 0.8)
prev = None
for s, e, name in rows[i0:i0 + 16]:
    print('%-70s %8.1f us  gap before %7.1f' % (name, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0)); prev = e
