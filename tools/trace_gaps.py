import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find the last band_tables_kernel -> next band_tables_kernel = one step
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('band_tables_kernel')]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = None
for r in rows[a - 3:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print('%9.1f us  dur %8.1f us  gap %7.1f us  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, r['Kernel_Name'].split('(')[0][-60:]))
    prev_end = max(prev_end or 0, e)
