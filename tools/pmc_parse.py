import csv, glob, collections, sys
d = sys.argv[1]
f = glob.glob(d + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "hint_block" in k:
        print(k)
        for c, x in sorted(v.items()):
            print("   %-28s %14.0f  (n=%d)" % (c, sum(x) / len(x), len(x)))
