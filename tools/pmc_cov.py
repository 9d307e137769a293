"""HBM bytes and duration of the covariance kernels of ONE fit from rocprofv3 passes (FETCH_SIZE, WRITE_SIZE in their own --pmc runs,
--kernel-trace in both): per kernel name the write / corrected read bytes (gfx950: FETCH_SIZE x 2, KiB units) and the time.
Usage: pmc_cov.py <dir with FETCH_SIZE/ and WRITE_SIZE/ sub-directories>"""
import csv, glob, json, os, sys
root = sys.argv[1]
out = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(root, counter, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            if "cov_" not in name or row.get("Counter_Name") != counter:
                continue
            k = name.replace("void gpmi::(anonymous namespace)::", "").split("(")[0]
            e = out.setdefault(k, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "ns": 0.0, "launches": 0})
            e[counter] += float(row["Counter_Value"]) * 1024.0
            if counter == "WRITE_SIZE":
                e["launches"] += 1
                try:
                    e["ns"] += float(row.get("End_Timestamp", 0)) - float(row.get("Start_Timestamp", 0))
                except ValueError:
                    pass
res = {}
for k, e in out.items():
    res[k] = {"launches": e["launches"], "write_bytes": e["WRITE_SIZE"], "read_bytes_corrected": 2.0 * e["FETCH_SIZE"], "ms": e["ns"] / 1e6,
              "write_GBps": (e["WRITE_SIZE"] / e["ns"]) if e["ns"] else None}
print(json.dumps(res, indent=1))
