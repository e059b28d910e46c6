"""Average counter values per launch of one kernel from a rocprofv3 --pmc run: python scripts/pmc_table.py <dir> <kernel substring>."""
import csv, glob, sys
from collections import defaultdict
d, flt = sys.argv[1], sys.argv[2]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc, n = defaultdict(float), defaultdict(int)
for f in files:
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]] += 1
for k in sorted(acc):
    print(f"{k:32s} {acc[k] / n[k]:14.4e}  ({n[k]} launches)")
