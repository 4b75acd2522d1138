"""Summarise a tools/profile_round.sh output directory: per-kernel stats and per-launch PMC averages."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(n):
    n = n.split('(')[0]
    return n.replace('eincm::', '').replace('void ', '')


for f in glob.glob(os.path.join(out, 'trace', '**', '*kernel_stats.csv'), recursive=True):
    print('== kernel stats (rocprofv3 --kernel-trace --stats):', os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(f"  {short(r['Name'])[:42]:42s} calls {r['Calls']:>6s} total_ns {r['TotalDurationNs']:>12s} avg_ns {float(r['AverageNs']):>12.1f} "
              f"pct {r['Percentage']:>6s}")

for tag in ('pmc_fetch', 'pmc_write', 'pmc_misc', 'pmc_valu'):
    for f in glob.glob(os.path.join(out, tag, '**', '*counter_collection.csv'), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
        print(f'== {tag}: per-launch average counter values:', os.path.relpath(f, out))
        for k, d in acc.items():
            for cn, v in d.items():
                print(f'  {k[:42]:42s} {cn:24s} n={len(v):5d} avg={sum(v) / len(v):.6g}')
