import csv, sys, collections, glob, os
root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "*/"))):
    f = os.path.join(d, "p_counter_collection.csv")
    if not os.path.exists(f): continue
    rows = list(csv.DictReader(open(f)))
    kt = {r["Dispatch_Id"]: r for r in csv.DictReader(open(os.path.join(d, "p_kernel_trace.csv")))}
    agg = collections.OrderedDict()
    for r in rows:
        name = r["Kernel_Name"][:60]
        grid = r.get("Grid_Size", "")
        key = (name, grid)
        a = agg.setdefault(key, collections.OrderedDict())
        c = a.setdefault(r["Counter_Name"], [0.0, 0])
        c[0] += float(r["Counter_Value"]); c[1] += 1
    print("==", os.path.basename(d.rstrip("/")))
    for (name, grid), cs in agg.items():
        if "conv_igemm" not in name: continue
        print("  %-62s grid=%-8s " % (name, grid) + "  ".join("%s=%.4g" % (k, v[0] / v[1]) for k, v in cs.items()))
