#!/bin/bash
# round 5, call R: does the trace build behave like the product?  (1) every chain launch of one blocked fit, one line each;
# (2) the blocked bench's chain phase measured on the TRACE library (copied over the product library on the box only), marks off
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
timeout 600 python tools/chain_trace.py --keep $O/chain_trace_raw --brief 1 --skip 0 --launches 19 20000 blocked --skip 0 --launches 12 50000 blocked
cp tools/bin/libgpmi_chain_trace.so gaussianprocesses.jl_amd/lib/libgpmi.so
echo "== blocked one rank on the trace library (marks off): per-step phases"
timeout 600 python bench.py --steps 3 --warmup 1 --mode sharded --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('  N=50000 ms/step %.1f' % j['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in j['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')}); c=j['c2']; print('  c2 ms/step %.2f' % c['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in c['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')})"
} > $O/r05_r_trace_vs_product.log 2>&1
cat $O/r05_r_trace_vs_product.log
