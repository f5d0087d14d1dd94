"""print the kernel timeline of the last step from a rocprofv3 kernel-trace csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last occurrence of the pair kernel -> walk back to the previous one
idx = [i for i, r in enumerate(rows) if 'k_pair_wave' in r['Kernel_Name'] and 'FamNbr' not in r['Kernel_Name']]
i1, i0 = idx[-1], idx[-2]
t0 = int(rows[i0]['End_Timestamp'])
prev_end = t0
for r in rows[i0 + 1:i1 + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%8.1f us  gap %6.1f  dur %7.1f  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r['Kernel_Name'][:90]))
    prev_end = e
print('step (pair end to pair end): %.1f us' % ((int(rows[i1]['End_Timestamp']) - t0) / 1e3))
