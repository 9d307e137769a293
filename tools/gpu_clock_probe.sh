#!/bin/bash
# average shader clock during the update kernel at K = 256 / 1024 and during the bare MFMA loop: GRBM_GUI_ACTIVE / duration
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/clk"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$O/run" -- python "$R/tools/gemm_phases.py" all 0 > "$O/run.log" 2>&1
cd "$R"
python - <<'PY'
import csv, glob, collections
cc = glob.glob("gpurun_out/clk/run/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob("gpurun_out/clk/run/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "")))
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    d, name, grid = dur.get(r["Dispatch_Id"], (0, "", ""))
    if d > 100000 and "gemm_nt" in name:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        print(f"{name[:60]:60s} grid {grid:>8s} dur {d/1e3:9.1f} us  clock {float(r['Counter_Value']) / 8 / d:.3f} GHz")
PY
rm -rf "$O"
