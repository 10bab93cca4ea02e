"""Summarise a rocprofv3 kernel trace: per conv launch (tiles, microseconds) of the last step."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
name = sys.argv[2] if len(sys.argv) > 2 else 'conv_tiles_kernel<9>'
n = int(sys.argv[3]) if len(sys.argv) > 3 else 105
sel = [r for r in rows if name in r['Kernel_Name']][-n:]
tot = 0
out = []
for r in sel:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    g = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
    tot += d
    out.append('%dx%sx%s %.1f' % (g, r['Grid_Size_Y'], r['Grid_Size_Z'], d))
print(' | '.join(out))
print('total us', round(tot, 1), 'vgpr', sel[-1]['VGPR_Count'], 'agpr', sel[-1]['Accum_VGPR_Count'], 'lds', sel[-1]['LDS_Block_Size'], 'scratch', sel[-1]['Scratch_Size'])
