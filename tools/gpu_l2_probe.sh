#!/bin/bash
# L2 behaviour of the update kernel at the bench shape: hit/miss counts and bytes to the fabric (separate --pmc passes).
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/l2_probe"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum" "FETCH_SIZE"; do
  tag=$(echo $c | tr ' ' '_')
  GPMI_GEMM_NI=4 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/$tag" -- python "$R/tools/gemm_phases.py" one 0 > "$O/$tag.log" 2>&1
  echo "pmc $tag exit $?"
  f=$(find "$O/$tag" -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_nt_kernel" not in r["Kernel_Name"]: continue
    a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in acc.items(): print(f"  {k}: {v/n:.4g} per launch ({n} launches)")
PY
done
