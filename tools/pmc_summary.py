"""Summarises rocprofv3 --pmc CSVs (gpurun_out/pmc_<tag>_*/pmc_counter_collection.csv) per kernel -> JSON on stdout."""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1]
out = {}
for d in sorted(glob.glob("gpurun_out/pmc_%s_*/pmc_counter_collection.csv" % tag)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d)):
        k = r["Kernel_Name"].split("(")[0]
        if "mpx" not in k:
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        out.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in cs.items()})
json.dump(out, sys.stdout, indent=1)
