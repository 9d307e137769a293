#!/bin/bash
# round 4: the A/B experiments behind profiles/r04_{e,f,g,h,i,m,o,q}_*.log, one section per argument (the per-call scripts they were
# run from are in the git history).  usage:  gpurun -- 'bash tools/gpu_r04_experiments.sh world1 fitc partitions'
mkdir -p gpurun_out; O=gpurun_out
line() { python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$1', 'ms %.1f fit %.1f predict %.1f frac %.3f' % (j['ms_per_step'], j['fit_only_ms_per_step'], j['predict_only_ms_per_step'], j['roofline']['frac']))"; }
for what in "$@"; do case $what in
world1)      # blocked handle on one rank against the dense path (r04_e_blocked_world1.log, r04_g: block widths at N = 20 000)
  for n in 50000 20000; do for mode in "" "--mode sharded"; do
    timeout 600 python bench.py --n $n --steps 5 --warmup 2 --no-cpu-baseline --no-secondary $mode 2>/dev/null | line "n=$n $mode"; done; done
  for wd in 512 1024 2048; do GPMI_DIST_WD=$wd timeout 300 python bench.py --n 20000 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --mode sharded 2>/dev/null | line "n=20000 WD=$wd"; done ;;
fitc)        # FITC C5 with the tall products on update256_kernel against the 128 x 128 kernel (r04_f_fitc_c5.log)
  for u in 1 0; do GPMI_UPDATE256=$u timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --secondary c5 2>/dev/null | python -c "
import sys, json
c = json.loads(sys.stdin.readline())['c5']
print('update256=$u c5: fit %.3f s (%.3f of peak) grad %.3f s parity' % (c['update_mll_s'], c['update_mll_frac_of_fp64_matrix_peak'], c['update_dmll_s']), c['parity']['rel_err'])"; done ;;
sweeps)      # GPMI_UPDATE256_MIN / GPMI_SUPER at the bench size (r04_h_sweeps.log)
  for cfg in "GPMI_UPDATE256_MIN=1024" "GPMI_UPDATE256_MIN=512" "GPMI_UPDATE256_MIN=256" "GPMI_SUPER=8192,12288,20480" "GPMI_SUPER=8192,16384,28672"; do
    env $cfg timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | line "$cfg"; done ;;
atomic)      # update256_kernel's atomic epilogue (r04_m_atomic.log)
  for a in 0 1; do GPMI_UPDATE256_ATOMIC=$a timeout 300 python tools/update256_vs_128.py f64 2>&1 | grep -v amdgpu | head -4
    GPMI_UPDATE256_ATOMIC=$a timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | line "atomic=$a"; done ;;
partitions)  # a model sharded over the two CU partitions, with 5 ms injected in front of the exchange / the broadcast (r04_i, r04_o, r04_q)
  cat > /tmp/part_fit.py <<'PY'
import math, os, sys, time
import numpy as np
root = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(root, "gaussianprocesses.jl_amd")); sys.path.insert(0, root)
import gpmi355x as g
from gpmi355x import dist as gd
n, d = int(sys.argv[1]), 8
rng = np.random.default_rng(17)
x = rng.uniform(size=(d, n)); y = np.sin(2 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
ll = [math.log(0.5) + 0.05 * k for k in range(d)]
gp = gd.ShardedGPE(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1), ctx=g.Context(devices=[256, 512]), block=1024)
def fit():
    ts = []
    for rep in range(3):
        t0 = time.perf_counter(); gp.update_mll(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
t0 = fit(); out = [f"no delay {t0:.0f}"]
for what, name in ((2, "exchange"), (1, "broadcast")):
    os.environ["GPMI_TEST_COMM_DELAY_ON"] = str(what); os.environ["GPMI_TEST_COMM_DELAY_US"] = "5000"
    t = fit(); out.append(f"{name}+5ms {t:.0f} (exposed {(t - t0) / (5.0 * (n // 1024 - (1 if what == 2 else 0))):.2f})")
print(f"two partitions n={n}: " + ", ".join(out) + f", mll {gp.mll:.4f}")
PY
  for n in 32768 65536; do timeout 300 python /tmp/part_fit.py $n 2>&1 | grep partitions; done ;;
esac; done 2>&1 | tee -a $O/r04_experiments.log
