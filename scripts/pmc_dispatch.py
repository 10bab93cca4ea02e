"""Per-dispatch view of a rocprofv3 --pmc (+ --kernel-trace) run: for the kernel whose name contains
argv[2], the dispatches with the largest grids, their counters and (if traced) duration.
usage: pmc_dispatch.py <dir> <name substring> [rows]"""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if sys.argv[2] not in r['Kernel_Name']:
        continue
    e = disp.setdefault(r['Dispatch_Id'], {'grid': int(r['Grid_Size']), 'wg': int(r['Workgroup_Size'])})
    e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
dur = {}
tf = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
if tf:
    for r in csv.DictReader(open(tf[0])):
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
rows = sorted(disp.items(), key=lambda kv: -kv[1]['grid'])[:int(sys.argv[3]) if len(sys.argv) > 3 else 6]
for k, e in rows:
    print('dispatch %s workgroups %d dur_us %s' % (k, e['grid'] // e['wg'], dur.get(k)))
    for n, v in sorted(e.items()):
        if n not in ('grid', 'wg'):
            print('    %-32s %.6g' % (n, v))
