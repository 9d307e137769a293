"""L2 counters of the trailing-update class from two --pmc passes over the bench command (tools/gpu_r05_l.sh): hit rate of the XCDs' L2s and the
read requests they receive, per launch.  Prints one JSON object."""
import csv, glob, json, os, sys

root = sys.argv[1]
KERNELS = ("gemm_nt_kernel<double, 0, 4>", "update256_kernel<double, 0")
out = {"kernel": " + ".join(KERNELS), "command": "python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary", "units": "per launch (summed over the 8 XCDs)"}
for folder in ("tcc", "tcp"):
    per = {}
    for f in glob.glob(os.path.join(root, folder, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if any(k in row.get("Kernel_Name", "") for k in KERNELS):
                key = (row["Counter_Name"], row["Dispatch_Id"])
                per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
    names = sorted({k[0] for k in per})
    for n in names:
        v = [val for (c, d), val in per.items() if c == n]
        out[n + "_avg"] = sum(v) / len(v)
        out[n + "_launches"] = len(v)
if out.get("TCC_HIT_sum_avg") and out.get("TCC_REQ_sum_avg"):
    out["l2_hit_rate_all_requests"] = out["TCC_HIT_sum_avg"] / out["TCC_REQ_sum_avg"]
if out.get("TCP_TCC_READ_REQ_sum_avg"):
    out["l2_read_bytes_at_128B_per_request"] = out["TCP_TCC_READ_REQ_sum_avg"] * 128.0
print(json.dumps(out))
