"""Per-launch averages of the K1 counters collected by tools/pmc_k1_r2.sh -> JSON on stdout."""
import csv, glob, json, os, sys
out = {}
for sub in "abc":
    fs = glob.glob(f"{sys.argv[1]}/{sub}/*/*_counter_collection.csv")
    if not fs:
        continue
    d = {}
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):      # the newest run (files are named by pid)
        kn = r["Kernel_Name"]
        if kn.startswith(("void k1p_loglikes", "k1p_loglikes", "void k1_loglikes", "k1_loglikes", "void k1b_loglikes", "k1b_loglikes", "void k1h_loglikes", "k1h_loglikes", "void k1s_loglikes", "k1s_loglikes")):
            d.setdefault(kn.split("(")[0], {}).setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            d[kn.split("(")[0]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for kn, cs in d.items():
        for c, v in cs.items():
            out.setdefault(kn, {})[c] = sum(v.values()) / len(v)
print(json.dumps(out, indent=1))
