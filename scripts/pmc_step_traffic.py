"""HBM bytes moved by ONE train step, from two rocprofv3 --pmc passes over bench.py (FETCH_SIZE in
one, WRITE_SIZE in the other; KB).  A step = the dispatches between the last two launches of
adam_tensors_kernel.  Reads = 2 x FETCH_SIZE (gfx950 tallies 128-byte requests of wide reads as 64).
usage: pmc_step_traffic.py <fetch dir> <write dir> <out json>"""
import csv, glob, json, sys, collections

def step_sums(d, counter):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        e = disp.setdefault(int(r['Dispatch_Id']), [r['Kernel_Name'], 0.0])
        e[1] += float(r['Counter_Value'])
    ids = sorted(disp)
    marks = [i for i in ids if 'adam_tensors_kernel' in disp[i][0]]
    lo, hi = marks[-2], marks[-1]
    per = collections.defaultdict(float)
    for i in ids:
        if lo < i <= hi:
            per[disp[i][0].split('(')[0][:48]] += disp[i][1] * 1024
    return per

rd, wr = step_sums(sys.argv[1], 'FETCH_SIZE'), step_sums(sys.argv[2], 'WRITE_SIZE')
kernels = {k: {'read_bytes': 2 * rd.get(k, 0.0), 'write_bytes': wr.get(k, 0.0)} for k in set(rd) | set(wr)}
top = dict(sorted(kernels.items(), key=lambda kv: -(kv[1]['read_bytes'] + kv[1]['write_bytes']))[:12])
out = {'read_bytes_per_step': 2 * sum(rd.values()), 'write_bytes_per_step': sum(wr.values()),
       'hbm_bytes_per_step': 2 * sum(rd.values()) + sum(wr.values()), 'largest_kernels': top,
       'method': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs of bench.py --steps 2 --warmup 1; '
                 'dispatches between the last two adam_tensors_kernel launches; reads = 2 x FETCH_SIZE (gfx950)'}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print('step: read %.2f GB, written %.2f GB' % (out['read_bytes_per_step'] / 1e9, out['write_bytes_per_step'] / 1e9))
