"""Dev tool: per-kernel means of the counters in rocprofv3 --pmc CSV outputs.  argv: <counter_collection.csv> [name filter]"""
import collections, csv, sys
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if flt and flt not in n:
        continue
    acc[n[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in acc.items():
    print(n, " dispatches:", len(next(iter(cs.values()))))
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v) / len(v):16.0f}")
