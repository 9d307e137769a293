"""Per-launch HBM traffic of the trailing-update kernel from the two --pmc passes of tools/gpu_pmc_bench.sh.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 128-B read requests at 64 B
(MI355X_MICROARCH.md, HBM section), so reads are doubled.  Prints one JSON object."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
KERNELS = ("gemm_nt_kernel<double, 0, 4>", "update256_kernel<double, 0")  # the trailing update: 128 x 128 and 256 x 128 forms (one roofline class;
# "<double, 0": not the <double, 15, ...> launches of gpmi_mfma_peak, the bare MFMA stream bench.py samples before and after the timed steps)
KERNEL = " + ".join(KERNELS)
out = {"commit": os.environ.get("GPMI_COMMIT", "unknown"), "kernel": KERNEL, "command": "python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary", "units": "bytes per launch"}
for counter, folder in (("FETCH_SIZE", "FETCH_SIZE"), ("WRITE_SIZE", "WRITE_SIZE"), ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES"),
                        ("SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES"), ("GRBM_GUI_ACTIVE", "GRBM_GUI_ACTIVE")):
    vals = {}
    for f in glob.glob(os.path.join(root, folder, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if any(k in row.get("Kernel_Name", "") for k in KERNELS) and row.get("Counter_Name") == counter:
                key = row.get("Dispatch_Id")
                vals[key] = vals.get(key, 0.0) + float(row["Counter_Value"])
    n = len(vals)
    out[counter + "_launches"] = n
    out[counter + ("_KiB_avg" if counter.endswith("_SIZE") else "_avg")] = (sum(vals.values()) / n) if n else None
if out.get("FETCH_SIZE_KiB_avg") and out.get("WRITE_SIZE_KiB_avg"):
    rd = 2.0 * out["FETCH_SIZE_KiB_avg"] * 1024.0
    wr = out["WRITE_SIZE_KiB_avg"] * 1024.0
    out["read_bytes_corrected"] = rd
    out["write_bytes"] = wr
    out["traffic_bytes_per_launch"] = rd + wr
# the kernel build (cov!): bytes of the covariance kernels per fit + predict (the N x N assembly by cov_fast_kernel + cov_kernel
# and the K*' rows of predict), for the HBM GB/s of the memory-bound stage; fits = big cov_fast launches
cov = {}
fits = 0
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    tot = 0.0
    for f in glob.glob(os.path.join(root, counter, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            if ("cov_fast_kernel" in name or "cov_kernel<" in name) and row.get("Counter_Name") == counter:
                v = float(row["Counter_Value"]) * 1024.0
                tot += v
                if counter == "WRITE_SIZE" and "cov_fast_kernel" in name and v > 2 ** 30:
                    fits += 1
    cov[counter] = tot
if fits:
    out["cov_fits"] = fits
    out["cov_write_bytes_per_fit"] = cov["WRITE_SIZE"] / fits
    out["cov_read_bytes_per_fit_corrected"] = 2.0 * cov["FETCH_SIZE"] / fits
if out.get("SQ_VALU_MFMA_BUSY_CYCLES_avg") and out.get("GRBM_GUI_ACTIVE_avg"):
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; the chip has 256 CUs x 4 SIMDs whose MFMA pipes each count busy cycles
    cycles = out["GRBM_GUI_ACTIVE_avg"] / 8.0
    out["mfma_busy_frac"] = out["SQ_VALU_MFMA_BUSY_CYCLES_avg"] / (1024.0 * cycles)
    if out.get("SQ_INSTS_MFMA_avg"):
        out["mfma_busy_cycles_per_instruction"] = out["SQ_VALU_MFMA_BUSY_CYCLES_avg"] / out["SQ_INSTS_MFMA_avg"]
print(json.dumps(out))
