set -u
O=gpurun_out/r03ae
mkdir -p $O
for L in 256 224 192 160 256; do
  export BBDM_BF3P_CUS=$L
  ( timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu --no-f32mfma --no-parity > $O/bench_c2_$L.json 2> $O/bench_c2_$L.err )
  python -c "
import json; d=json.load(open('$O/bench_c2_$L.json')); print('cus$L', round(d['ms_per_step'],3), {k[:28]: round(v,2) for k,v in sorted(d['kernel_ms_per_step'].items(), key=lambda kv:-kv[1]) if v > 5.0})"
done
