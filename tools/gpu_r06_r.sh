#!/bin/bash
# round 6, call R: strip height of the tile order (tile_order.h TILE_GROUP: 8 tile-rows per strip since round 2) — A/B libraries built by
# tools/build_variant.sh with -DGPMI_TILE_GROUP=4 / 16 against the product library, on the dense bench line and C2.
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
L=gaussianprocesses.jl_amd/lib/libgpmi.so
cp $L /tmp/libgpmi_base.so
{
for v in base tg4 tg16 base tg16; do
if [ $v = base ]; then cp /tmp/libgpmi_base.so $L; else cp tools/bin/libgpmi_$v.so $L; fi
echo "== dense, library $v"
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 > $O/r06_r_bench_$v.json
python -c "import json; j=json.load(open('$O/r06_r_bench_$v.json')); r=j['roofline']; print('  N=50000 ms/step %.1f fits/s %.4f frac %.3f fit %.1f predict %.1f; c2 %.2f fit %.2f predict %.2f frac %.3f' % (j['ms_per_step'], j['value'], r['frac'], j.get('fit_only_ms_per_step',0), j.get('predict_only_ms_per_step',0), j['c2']['ms_per_step'], j['c2']['fit_only_ms_per_step'], j['c2']['predict_only_ms_per_step'], j['c2']['roofline_frac']))"
done
cp /tmp/libgpmi_base.so $L
} > $O/r06_r.log 2>&1
cat $O/r06_r.log
