"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: sum of every counter and dispatch count."""
import csv, sys, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for path in sys.argv[1:]:
    for f in glob.glob(path, recursive=True):
        seen = set()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (row["Dispatch_Id"], k)
            if key not in seen:
                seen.add(key); cnt[k] += 1
for k in sorted(agg, key=lambda k: -cnt[k]):
    if not k.startswith("void k_") and not k.startswith("k_"): continue
    print(k, "dispatches", cnt[k])
    for c, v in sorted(agg[k].items()):
        print("   %-32s %.4g" % (c, v))
