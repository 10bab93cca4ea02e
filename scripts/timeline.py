"""Timeline of the last full step in a rocprofv3 kernel trace (CSV): start, gap to the previous
kernel's end, duration, name, workgroups.  usage: timeline.py <trace dir> <out file>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a step ends with the optimizer's run of multi-tensor launches
idx = [i for i, r in enumerate(rows) if 'FusedOptimizer' in r['Kernel_Name'] or 'adam_tensors_kernel' in r['Kernel_Name']]
ends = [i for j, i in enumerate(idx) if j + 1 == len(idx) or idx[j + 1] != i + 1]
s, e = ends[-2] + 1, ends[-1] + 1
t0 = int(rows[s]['Start_Timestamp'])
prev_end = t0
out = []
gaps = 0.0
for r in rows[s:e]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (st - prev_end) / 1e3
    gaps += max(gap, 0)
    wg = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']) * int(r['Grid_Size_Y'])
    out.append('%8.1f gap %6.1f dur %7.1f  %-50s wg %d' % ((st - t0) / 1e3, gap, (en - st) / 1e3, r['Kernel_Name'][:50], wg))
    prev_end = max(prev_end, en)
out.append('step span %.1f us, idle gaps %.1f us' % ((prev_end - t0) / 1e3, gaps))
open(sys.argv[2], 'w').write('\n'.join(out) + '\n')
