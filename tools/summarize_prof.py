"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary per kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, '**', pattern), recursive=True))


for f in find('*kernel_stats.csv'):
    print('== kernel stats:', f)
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print(f"{r.get('Name', '')[:70]:70s} calls={r.get('Calls')} avg_ns={r.get('AverageNs')} total_ns={r.get('TotalDurationNs')} pct={r.get('Percentage')}")

for f in find('*counter_collection.csv'):
    print('== counters:', f)
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    for r in csv.DictReader(open(f)):
        pass
    disp = defaultdict(set)
    for r in csv.DictReader(open(f)):
        disp[r['Kernel_Name'][:60]].add(r['Dispatch_Id'])
    for k, d in acc.items():
        n = max(1, len(disp[k]))
        print(f'{k:60s} dispatches={n} ' + ' '.join(f'{c}={v / n:.4g}' for c, v in sorted(d.items())))
