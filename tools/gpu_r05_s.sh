#!/bin/bash
# round 5, call S (run twice: the second time with the shared-fragment product64): the timeline again with the wait clock only in the slow path of a wait (s_memrealtime costs ~1 us: with two reads
# around EVERY flag check, call Q trace build ran the K loops of tasks ahead of the completed-column mark at 8 us per slab)
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
timeout 600 python tools/chain_trace.py --keep $O/chain_trace_raw --skip 3 --launches 1 20000 blocked 50000 blocked --skip 0 --launches 1 20000 blocked
cp tools/bin/libgpmi_chain_trace.so gaussianprocesses.jl_amd/lib/libgpmi.so
echo "== blocked one rank on the trace library (marks off): per-step phases"
timeout 600 python bench.py --steps 3 --warmup 1 --mode sharded --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('  N=50000 ms/step %.1f' % j['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in j['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')}); c=j['c2']; print('  c2 ms/step %.2f' % c['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in c['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')})"
} > $O/r05_s2_chain_trace_product64.log 2>&1
cat $O/r05_s2_chain_trace_product64.log
