"""Per-launch averages of the issue-level counters (passes a/b/c of tools/profile_r3.sh) for the step's kernels -- K1, the K2 DP and
the K3 accumulate kernels -- with a few derived ratios -> JSON on stdout."""
import csv, glob, json, os, sys
out = {}
for sub in "abc":
    fs = glob.glob(f"{sys.argv[1]}/{sub}/*/*_counter_collection.csv")
    if not fs:
        continue
    d = {}
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):      # the newest run (files are named by pid)
        kn = r["Kernel_Name"]
        if kn.startswith(("void k1p_loglikes", "k1p_loglikes", "void k1_loglikes", "k1_loglikes", "void k1b_loglikes", "k1b_loglikes", "void k1h_loglikes", "k1h_loglikes", "void k1s_loglikes", "k1s_loglikes", "void k2_viterbi_dp", "k2_viterbi_dp", "void k3_accumulate", "k3_accumulate")):
            d.setdefault(kn.split("(")[0], {}).setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            d[kn.split("(")[0]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for kn, cs in d.items():
        for c, v in cs.items():
            out.setdefault(kn, {})[c] = sum(v.values()) / len(v)
for kn, m in out.items():
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        d = {}
        if "SQ_ACTIVE_INST_VALU" in m: d["valu_active_per_wave_cycle"] = m["SQ_ACTIVE_INST_VALU"] / wc
        if "SQ_WAIT_ANY" in m: d["wait_any_per_wave_cycle"] = m["SQ_WAIT_ANY"] / wc
        if "SQ_WAIT_INST_ANY" in m: d["wait_inst_any_per_wave_cycle"] = m["SQ_WAIT_INST_ANY"] / wc
        if "SQ_INSTS_VALU" in m and m.get("SQ_WAVES"): d["valu_insts_per_wave"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"]
        if "SQ_INSTS_SALU" in m and m.get("SQ_WAVES"): d["salu_insts_per_wave"] = m["SQ_INSTS_SALU"] / m["SQ_WAVES"]
        if "GRBM_GUI_ACTIVE" in m and "SQ_INSTS_VALU" in m:      # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs, 4 cycles per wave64 VALU instruction
            d["valu_issue_cycles_over_simd_cycles"] = 4.0 * m["SQ_INSTS_VALU"] / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0)
        m["derived"] = d
print(json.dumps(out, indent=1))
