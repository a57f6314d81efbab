set -u
O=gpurun_out/r03bb
mkdir -p $O
( timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "winograd or wino" > $O/pytest_wino.txt 2>&1 ); tail -4 $O/pytest_wino.txt
( timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu --no-f32mfma > $O/bench_c2.json 2> $O/bench_c2.err )
python -c "
import json; d=json.load(open('$O/bench_c2.json')); print('c2', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],3), 'frac_step', round(d['roofline']['frac_step'],3), {k[5:24]: round(v,2) for k,v in sorted(d['kernel_ms_per_step'].items(), key=lambda kv:-kv[1]) if v > 0.2}, d.get('parity'))" || tail -5 $O/bench_c2.err
for w in c3 c5 c1; do ( timeout 300 python bench.py --workload $w --no-cpu --no-pipeline > $O/bench_$w.json 2> $O/bench_$w.err ); python -c "
import json; d=json.load(open('$O/bench_$w.json')); print('$w', round(d['ms_per_step'],3), (d.get('parity') or {}).get('rel_err_x0_recon'))" || tail -3 $O/bench_$w.err; done
