"""Per-kernel averages of rocprofv3 PMC counters (counter_collection.csv)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-40:]
    c = r["Counter_Name"]
    a = acc[k][c]
    a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    print(k, {c: (round(v[0] / v[1], 1), v[1]) for c, v in cs.items()})
