set -u
O=gpurun_out/r03bc
mkdir -p $O
( timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "winograd or wino" > $O/pytest_wino.txt 2>&1 ); tail -3 $O/pytest_wino.txt
for C in 1 2 4 8 16; do
  ( BBDM_WINO_INPUT_CPW=$C timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu --no-f32mfma --no-parity > $O/bench_c2_cpw$C.json 2> $O/bench_c2_cpw$C.err )
  python -c "
import json; d=json.load(open('$O/bench_c2_cpw$C.json')); print('cpw$C', round(d['ms_per_step'],3), 'frac_step', round(d['roofline']['frac_step'],3), {k[5:24]: round(v,2) for k,v in sorted(d['kernel_ms_per_step'].items(), key=lambda kv:-kv[1]) if v > 5}, d.get('parity'))" || tail -5 $O/bench_c2_cpw$C.err
done
