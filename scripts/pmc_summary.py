"""Sum rocprofv3 --pmc counters per kernel name (diagnostic)."""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[(k, r['Counter_Name'])] += 1
for k, d in acc.items():
    if 'conv_tiles' in k or 'wgrad' in k or 'gemm' in k or (len(sys.argv) > 2 and sys.argv[2] in k):
        print(k)
        for n, v in sorted(d.items()):
            print('   %-32s %.4g  (dispatches %d)' % (n, v, cnt[(k, n)]))
