import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(float); ndisp = collections.defaultdict(int)
for f in glob.glob(os.path.join(out, "p*", "*", "*counter_collection.csv")):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
for f in glob.glob(os.path.join(out, "p1", "*", "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6; ndisp[k] += 1
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for k in sorted(agg, key=lambda k: -dur.get(k, 0)):
        line = "%-40s disp=%d ms=%.3f " % (k[:40], ndisp.get(k, 0), dur.get(k, 0)) + " ".join(
            "%s=%.4g" % (c, v) for c, v in sorted(agg[k].items()))
        fo.write(line + "\n")
        if dur.get(k, 0) > 0.3: print(line)
