#!/usr/bin/env python3
"""Summarises a rocprofv3 --pmc counter_collection.csv per kernel (mean over dispatches)."""
import csv, collections, sys, glob
path = sys.argv[1]
files = glob.glob(path + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in files:
    for r in csv.DictReader(open(fn)):
        name = r['Kernel_Name'].split('(')[0]
        if name.startswith('void '): name = name[5:]      # template instances: "void k_describe<0>(...)"
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    if not k.startswith('k_'):
        continue
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())}, 'dispatches', len(next(iter(d.values()))))
