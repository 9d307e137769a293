"""Instruction-side roofline of the covariance kernels from rocprofv3 --pmc passes (one sub-directory per pass under <root>):
per kernel name the counters summed over its launches, VALU lane-instructions per matrix entry written, VALU issue utilisation.
Usage: pmc_cov_valu.py <root> <entries written by the big launch, e.g. n*(n+1)/2>"""
import csv, glob, json, os, sys
root, entries = sys.argv[1], float(sys.argv[2])
acc = {}
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if "cov_" not in name:
            continue
        k = name.replace("void gpmi::(anonymous namespace)::", "").split("(")[0]
        e = acc.setdefault(k, {})
        c = row["Counter_Name"]
        e[c] = e.get(c, 0.0) + float(row["Counter_Value"])
        if c == "SQ_WAVES":
            e["_ns"] = e.get("_ns", 0.0) + float(row.get("End_Timestamp", 0)) - float(row.get("Start_Timestamp", 0))
            e["_launches"] = e.get("_launches", 0) + 1
            e["_vgpr"] = row.get("VGPR_Count") or row.get("Arch_VGPR_Count")
res = {}
for k, e in acc.items():
    r = {c: v for c, v in e.items() if not c.startswith("_")}
    r["launches"], r["ms_under_pmc"], r["vgprs"] = e.get("_launches"), e.get("_ns", 0.0) / 1e6, e.get("_vgpr")
    if e.get("SQ_INSTS_VALU"):
        r["valu_lane_instructions_per_entry"] = 64.0 * e["SQ_INSTS_VALU"] / entries
    if e.get("SQ_ACTIVE_INST_VALU") and e.get("SQ_BUSY_CYCLES"):
        # SQ_ACTIVE_INST_VALU: cycles (x4, per SIMD quad-cycle convention) a wave spent issuing VALU; SQ_BUSY_CYCLES summed over SEs
        r["valu_active_over_busy"] = e["SQ_ACTIVE_INST_VALU"] / e["SQ_BUSY_CYCLES"]
    if e.get("SQ_WAVE_CYCLES") and e.get("SQ_BUSY_CYCLES"):
        r["mean_waves_resident"] = e["SQ_WAVE_CYCLES"] / e["SQ_BUSY_CYCLES"]
    res[k] = r
print(json.dumps(res, indent=1))
