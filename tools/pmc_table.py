"""Mean per launch of every counter in a rocprofv3 --pmc csv, per kernel: python tools/pmc_table.py <dir> [kernel substring]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            k = (r["Kernel_Name"][:48], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
for (kn, cn), (s, n) in sorted(acc.items()):
    print("%-48s %-28s mean %.4g  (n=%d)" % (kn, cn, s / n, n))
