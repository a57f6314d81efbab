set -u
O=gpurun_out/r03ad
mkdir -p $O
for L in remap base remap base; do
  if [ $L = base ]; then unset BBDM_HIP_LIB; else export BBDM_HIP_LIB=$PWD/tools/_nt_$L.so; fi
  ( timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu --no-f32mfma > $O/bench_c2_$L.json 2> $O/bench_c2_$L.err )
  python -c "
import json; d=json.load(open('$O/bench_c2_$L.json')); print('$L', round(d['ms_per_step'],3), {k:v for k,v in d['parity'].items() if k.startswith('rel')}, {k[:28]: round(v,2) for k,v in sorted(d['kernel_ms_per_step'].items(), key=lambda kv:-kv[1]) if v > 1.0})"
done
