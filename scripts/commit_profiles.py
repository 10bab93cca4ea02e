"""Copies what scripts/refresh_profiles.sh left under gpurun_out/ into profiles/ under the round's
names and stamps the counter summaries with the commit they were measured at.
usage: python scripts/commit_profiles.py r2"""
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, 'gpurun_out'), os.path.join(root, 'profiles')
head = subprocess.check_output(['git', '-C', root, 'rev-parse', '--short', 'HEAD'], text=True).strip()
names = {'prof_bench.json': 'bench_n1.json', 'prof_bench_under_rocprof.json': 'bench_n1_under_rocprof.json',
         'prof_kernel_stats.csv': 'bench_n1_kernel_stats.csv', 'prof_timeline.txt': 'step_timeline_under_rocprof.txt',
         'prof_pmc_summary.txt': 'pmc_summary.txt', 'prof_pmc_resblock_traffic.json': 'pmc_resblock_traffic.json',
         'prof_pmc_step_traffic.json': 'pmc_step_traffic.json'}
for a, b in names.items():
    out = os.path.join(dst, '%s_%s' % (tag, b))
    if b.startswith('pmc_') and b.endswith('.json'):
        with open(os.path.join(src, a)) as f:
            j = json.load(f)
        j['commit'] = head
        with open(out, 'w') as f:
            json.dump(j, f, indent=1)
    else:
        shutil.copyfile(os.path.join(src, a), out)
    print(out)
