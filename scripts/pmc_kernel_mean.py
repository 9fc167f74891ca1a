"""mean of each PMC counter over the dispatches of one kernel:  pmc_kernel_mean.py KERNEL_SUBSTRING dir..."""
import csv, glob, sys, collections
KERNEL = sys.argv[1]
for d in sys.argv[2:]:
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r['Kernel_Name']:
                a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    print(d, {k: (v[0] / v[1], v[1]) for k, v in acc.items()})
