#!/bin/bash
# round 6, call S (the code it measures was REMOVED after this run — slower, and one bit-identity test red; log: profiles/r06_s_*): the diagonal task recomputing L(c, c-1) itself (chain_block_kernel<T, FOLD = true>, GPMI_CHAIN_FOLD = 0 | 1 | 2): same bits?
# then the chain tests under it, the dense line / C2 and the blocked handle.
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
echo "== same bits with and without"
timeout 300 python - <<'PY'
import math, os, sys
sys.path.insert(0, "gaussianprocesses.jl_amd"); sys.path.insert(0, ".")
import numpy as np
import gpmi355x as g
from oracle import gp_oracle as G
x, y, xs = G.synthetic_inputs(4100, 4, p=8)
spec = ("sum", ("se_ard", [math.log(0.4), math.log(0.5), math.log(0.6), math.log(0.7)], 0.0), ("mat32_iso", math.log(0.8), -0.5))
out = {}
for fold in ("0", "1"):
    os.environ["GPMI_CHAIN_FOLD"] = fold
    ctx = g.Context(0)
    gp = g.GP(x, y, g.MeanZero(), g.from_spec(spec), math.log(0.15), ctx=ctx)
    out[fold] = (gp.mll, np.triu(gp.cK.cholfactors()))
    del gp; ctx.close()
print("mll equal:", out["0"][0] == out["1"][0], "factor bit-identical:", np.array_equal(out["0"][1], out["1"][1]), out["0"][0])
PY
echo "== chain tests, GPMI_CHAIN_FOLD=1"
GPMI_CHAIN_FOLD=1 timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_twolevel.py -m gpu -x -q 2>&1 | tail -3
for fd in 0 1 2 0 2; do
echo "== dense, GPMI_CHAIN_FOLD=$fd"
GPMI_CHAIN_FOLD=$fd timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 > $O/r06_s_bench_dense_fold$fd.json
python -c "import json; j=json.load(open('$O/r06_s_bench_dense_fold$fd.json')); r=j['roofline']; print('  N=50000 ms/step %.1f fits/s %.4f frac %.3f fit %.1f; c2 %.2f fit %.2f frac %.3f' % (j['ms_per_step'], j['value'], r['frac'], j.get('fit_only_ms_per_step',0), j['c2']['ms_per_step'], j['c2']['fit_only_ms_per_step'], j['c2']['roofline_frac']))"
done
for fd in 0 1; do
echo "== blocked one rank, GPMI_CHAIN_FOLD=$fd"
GPMI_CHAIN_FOLD=$fd timeout 300 python bench.py --steps 3 --warmup 1 --mode sharded --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 > $O/r06_s_bench_blocked_fold$fd.json
python -c "
import json; j=json.load(open('$O/r06_s_bench_blocked_fold$fd.json')); print('  N=50000 ms/step %.1f' % j['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in j['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')}); c=j['c2']; print('  c2 ms/step %.2f' % c['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in c['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')})"
done
} > $O/r06_s.log 2>&1
cat $O/r06_s.log
