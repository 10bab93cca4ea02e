"""Average HBM traffic per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE in one,
WRITE_SIZE in the other; both in KB).  On gfx950 FETCH_SIZE counts a 128-byte request of a wide
streaming read as 64 bytes, so reads = 2 x FETCH_SIZE; WRITE_SIZE is exact.
usage: pmc_traffic.py <fetch dir> <write dir> <kernel name substring> <out json>"""
import csv, glob, json, sys

def per_dispatch(d, counter, name):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if name in r['Kernel_Name'] and r['Counter_Name'] == counter:
            acc[r['Dispatch_Id']] = acc.get(r['Dispatch_Id'], 0.0) + float(r['Counter_Value'])
    return acc

fetch = per_dispatch(sys.argv[1], 'FETCH_SIZE', sys.argv[3])
write = per_dispatch(sys.argv[2], 'WRITE_SIZE', sys.argv[3])
n = len(fetch)
assert n > 0 and n == len(write), (n, len(write))
rd = 2.0 * sum(fetch.values()) * 1024 / n
wr = sum(write.values()) * 1024 / n
json.dump({'kernel': sys.argv[3], 'launches': n, 'read_bytes_per_launch': rd, 'write_bytes_per_launch': wr,
           'hbm_bytes_per_launch': rd + wr,
           'method': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs of bench.py --steps 2 --warmup 1; '
                     'reads = 2 x FETCH_SIZE (gfx950), mean over every launch of the kernel'},
          open(sys.argv[4], 'w'), indent=1)
print(open(sys.argv[4]).read())
