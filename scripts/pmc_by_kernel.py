"""Sum rocprofv3 --pmc counters per kernel family: python scripts/pmc_by_kernel.py <dir with *counter_collection.csv> [name filter ...]"""
import collections, csv, os, sys
tot, n = collections.defaultdict(collections.Counter), collections.Counter()
for d, _, files in os.walk(sys.argv[1]):
    for f in files:
        if f.endswith("counter_collection.csv"):
            for r in csv.DictReader(open(os.path.join(d, f))):
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("rvb::", "")[:70]
                if len(sys.argv) > 2 and not any(w in k for w in sys.argv[2:]):
                    continue
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                n[(k, r["Counter_Name"])] += 1
for k in sorted(tot, key=lambda k: -sum(tot[k].values())):
    print(k, " ".join(f"{c}={v:.4g}(n={n[(k, c)]})" for c, v in sorted(tot[k].items())))
