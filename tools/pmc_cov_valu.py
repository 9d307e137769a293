"""Instruction-side roofline of the covariance kernels from rocprofv3 --pmc passes over ONE fit each (one sub-directory per pass
under <root>): per kernel name the counters PER FIT (a counter collected in several passes is averaged over them), VALU
lane-instructions per matrix entry written, VALU issue utilisation, resident waves.
Usage: pmc_cov_valu.py <root> [cycles a wave64 VALU instruction holds its SIMD: 4 for fp64 (default), 2 for fp32]"""
import csv, glob, json, os, sys
root = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    pas = os.path.relpath(f, root).split(os.sep)[0]
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if "cov_" not in name and "scale_inputs" not in name:
            continue
        k = name.replace("void gpmi::(anonymous namespace)::", "").split("(")[0]
        e = acc.setdefault(k, {})
        c = row["Counter_Name"]
        e.setdefault(c, {}).setdefault(pas, 0.0)
        e[c][pas] += float(row["Counter_Value"])
        if c == "SQ_WAVES":
            e.setdefault("_ns", {}).setdefault(pas, 0.0)
            e["_ns"][pas] += float(row.get("End_Timestamp", 0)) - float(row.get("Start_Timestamp", 0))
            e["_vgpr"] = row.get("VGPR_Count") or row.get("Arch_VGPR_Count")
res = {}
for k, e in acc.items():
    r = {c: sum(v.values()) / len(v) for c, v in e.items() if not c.startswith("_")}
    r["ms_under_pmc"] = sum(e["_ns"].values()) / len(e["_ns"]) / 1e6 if "_ns" in e else None
    r["vgprs"] = e.get("_vgpr")
    es = 4 if "<float" in k else 8
    cyc = 2.0 if es == 4 else 4.0
    if r.get("WRITE_SIZE") and r.get("SQ_INSTS_VALU"):
        entries = r["WRITE_SIZE"] * 1024.0 / es
        r["entries_written"] = entries
        r["valu_lane_instructions_per_entry"] = 64.0 * r["SQ_INSTS_VALU"] / entries
    if r.get("GRBM_GUI_ACTIVE"):
        cycles = r["GRBM_GUI_ACTIVE"] / 8.0                      # summed over the 8 XCDs
        r["kernel_cycles"] = cycles
        if r.get("ms_under_pmc"):
            r["effective_clock_GHz"] = cycles / (r["ms_under_pmc"] * 1e6)
        if r.get("SQ_INSTS_VALU"):
            r["valu_issue_utilisation"] = r["SQ_INSTS_VALU"] * cyc / (1024.0 * cycles)
        if r.get("SQ_WAVE_CYCLES"):
            r["mean_waves_per_simd"] = 4.0 * r["SQ_WAVE_CYCLES"] / cycles / 1024.0   # SQ_WAVE_CYCLES counts quad-cycles
    res[k] = r
print(json.dumps(res, indent=1))
