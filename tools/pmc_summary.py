import csv, glob, sys, collections
d = sys.argv[1]
for f in glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:60], r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, (c, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print("%-62s %-12s calls %7d  sum %.4g  mean/launch %.4g" % (k[0], k[1], c, v, v / c))
