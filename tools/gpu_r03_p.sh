#!/bin/bash
# round 3, call P5: the same bisect after the fix (attached events destroyed when the profile is drained), then the bench line from this tree
mkdir -p gpurun_out; O=gpurun_out
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu | tee $O/p_mem4.log
import ctypes as C, gc, sys, math
sys.path.insert(0, "gaussianprocesses.jl_amd"); sys.path.insert(0, ".")
import numpy as np
import bench
import gpmi355x as g
hip = C.CDLL("libamdhip64.so")
def free_gb():
    f, t = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value / 1e9
ctx = g.Context.default(0)
def report(tag, f0):
    gc.collect()
    print(f"{tag}: held after return {f0 - free_gb():.2f} GB", flush=True)
for (dt, steps, warm) in (("f32", 1, 0), ("f32", 2, 1), ("f64", 1, 0), ("f64", 5, 2)):
    f0 = free_gb()
    r = bench.run_workload(g, ctx, 30000, 16, 1024, dt, steps, warm, ctx.synchronize)
    del r
    report(f"run_workload {dt} steps={steps} warmup={warm}", f0)
x, y, xp = bench.synthetic_inputs(30000, 16, 1024)
def variant(name, body):
    f0 = free_gb()
    def run():
        gp = g.GP(x, y, g.MeanZero(), g.SEArd([0.0] * 16, 0.0), math.log(0.1), dtype=np.float32, ctx=ctx)
        body(gp)
    run()
    report(name, f0)
variant("create only", lambda gp: None)
variant("predict_f", lambda gp: gp.predict_f(xp))
variant("set_params + update_mll", lambda gp: (gp.set_params(np.asarray(gp.get_params()) + 1e-3), gp.update_mll()))
def prof(gp):
    ctx.profile_enable(True); gp.update_mll(); ctx.profile_get(g._lib.PROF_SYRK); ctx.profile_enable(False)
variant("profile_enable + update_mll", prof)
def closure(gp):
    w = {"a": 0.0}
    def step(i):
        gp.update_mll(); w["a"] += 1
        return gp.predict_f(xp)
    mu, s2 = step(0)
variant("closure step", closure)
PY
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "destroyed_model or synthetic_workload" 2>&1 | grep -v amdgpu | tail -3
timeout 900 python bench.py > $O/r_bench_full.json 2> $O/r_bench_full.err; echo "bench rc $?"; python - <<'PY'
import json
j = json.load(open("gpurun_out/r_bench_full.json"))
for k in ("ms_per_step", "fit_only_ms_per_step", "predict_only_ms_per_step", "value"):
    print(k, j[k])
print("roofline", {k: j["roofline"][k] for k in ("achieved", "frac", "traffic")})
print("parity", j.get("parity"))
for k in ("c2", "c4_single_gpu", "c3", "grad", "c5"):
    print(k, str(j.get(k))[:400])
print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["sample"][:120])
PY
