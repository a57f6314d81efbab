set -u
O=gpurun_out/r03ab
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "bf3p or winograd or phase" > $O/tests1.log 2>&1; echo "rc=$?" >> $O/tests1.log; tail -3 $O/tests1.log
for L in 3 2 3 2; do
  export BBDM_BF3P_STAGES=$L
  ( timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu --no-f32mfma > $O/bench_c2_$L.json 2> $O/bench_c2_$L.err )
  python -c "
import json; d=json.load(open('$O/bench_c2_$L.json')); print('stages$L', round(d['ms_per_step'],3), {k:v for k,v in d['parity'].items() if k.startswith('rel')}, {k[:28]: round(v,2) for k,v in sorted(d['kernel_ms_per_step'].items(), key=lambda kv:-kv[1]) if v > 1.0})"
done
