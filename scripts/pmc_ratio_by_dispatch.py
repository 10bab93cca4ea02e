"""Per dispatch, in dispatch order: kernel, workgroups and the ratio of two counters of a rocprofv3 --pmc run.
usage: pmc_ratio_by_dispatch.py <dir> <numerator counter> <denominator counter> [name substring]"""
import csv, glob, sys, collections
d, num, den = sys.argv[1:4]
sub = sys.argv[4] if len(sys.argv) > 4 else ''
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if sub not in r['Kernel_Name']:
        continue
    e = disp.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name'][:40], 'wg': int(r['Grid_Size']) // int(r['Workgroup_Size'])})
    e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
prev = None
for k in sorted(disp):
    e = disp[k]
    key = (e['name'], e['wg'], round(e.get(num, 0) / max(e.get(den, 1), 1), 4))
    if key != prev:
        print('dispatch %5d  %-40s wg %5d  %s %.4g  %s %.4g  ratio %.4f' % (k, e['name'], e['wg'], num, e.get(num, 0), den, e.get(den, 0), key[2]))
    prev = key
