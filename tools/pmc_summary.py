#!/usr/bin/env python3
"""summarise gpurun_out/pmc_<tag>/*/runc/*_counter_collection.csv: per kernel, mean counter value per dispatch"""
import csv, glob, sys, collections, re
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/pmc_{tag}/*/runc/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("yagi::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(f"gpurun_out/pmc_{tag}/trace/runc/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("yagi::", "")
        dur[k] = float(r["AverageNs"])
for k, cs in sorted(agg.items()):
    if "gen_kernel" in k or "update_window" in k or "rocclr" in k: continue
    print(f"== {k}  avg {dur.get(k, 0)/1e3:.1f} us")
    for c, v in sorted(cs.items()):
        # skip the first (cold) dispatch
        vv = v[1:] if len(v) > 1 else v
        print(f"   {c:28s} {sum(vv)/len(vv):16.0f}")
