set -u
O=gpurun_out/r03ac
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "winograd or phase or splitk" > $O/tests1.log 2>&1; echo "rc=$?" >> $O/tests1.log; tail -3 $O/tests1.log
for w in c3 c1 c5 c4; do
for L in 1 0; do
  export BBDM_WINO_OUTPUT_LDS=$L
  ( timeout 300 python bench.py --workload $w --no-cpu > $O/bench_${w}_$L.json 2> $O/bench_${w}_$L.err )
  python -c "
import json; d=json.load(open('$O/bench_${w}_$L.json')); print('$w lds$L', round(d['ms_per_step'],3), {k:(v if not isinstance(v,dict) else '') for k,v in (d['parity'] or {}).items() if k.startswith('rel')}, {k[:32]: round(v,2) for k,v in sorted((d.get('kernel_ms_per_step') or {}).items(), key=lambda kv:-kv[1])[:6]})"
done
done
